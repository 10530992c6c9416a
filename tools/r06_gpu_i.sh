#!/bin/bash
# round 6: BASELINE configs[3] / [4] on one GPU -- kernel trace, one FETCH / WRITE pass each, the bench line with roofline_workload + kernel_breakdown
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
for sc in dam_halfhalf_highres corner_dams_512; do
  bash tools/kstats_scene.sh $sc gpurun_out/r06_kernel_stats_$sc.csv 60
  head -25 gpurun_out/r06_kernel_stats_$sc.csv
  timeout 600 python bench.py --scene $sc --steps 60 --no-cpu-baseline --no-dense-pcg --no-fast-forward --no-other-schedule > gpurun_out/r06_bench_$sc.log 2>&1
  grep '^{' gpurun_out/r06_bench_$sc.log | tail -1 > gpurun_out/r06_bench_$sc.json
  python - <<P
import json
d=json.load(open("gpurun_out/r06_bench_$sc.json"))
print("$sc value", d["value"], "roofline_workload", d.get("roofline_workload"))
print(d["kernel_breakdown"]["us_per_step"], d["kernel_breakdown"]["sum_us_per_step"])
P
  cd /tmp && export TMPDIR=/tmp
  i=0
  for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1)); rm -rf $root/gpurun_out/_hp
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/_hp -o p -- python $root/bench.py --scene $sc --steps 20 --warmup 5 --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 --no-other-schedule > $root/gpurun_out/_hp.log 2>&1
    python $root/tools/pmc_summary.py $root/gpurun_out/_hp > $root/gpurun_out/r06_pmc_${sc}_$i.csv
  done
  rm -rf $root/gpurun_out/_hp
  cd $root
  head -12 gpurun_out/r06_pmc_${sc}_1.csv
done
