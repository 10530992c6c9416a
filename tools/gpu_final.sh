#!/bin/bash
# Final measurement pass of a round on the GPU box: tests, bench line, kernel trace, PMC captures, SQ counters, M4, slab loopback.
tag=${1:-r04}; ver=${2:-v1}
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/${tag}
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 -s > ${o}_gpu_tests_${ver}.log 2>&1; tail -3 ${o}_gpu_tests_${ver}.log
python -c "import __graft_entry__ as g; g.smoke()" > ${o}_smoke_${ver}.log 2>&1; tail -1 ${o}_smoke_${ver}.log
timeout 600 python bench.py > ${o}_bench_${ver}.log 2>&1; grep '^{' ${o}_bench_${ver}.log | tail -1 > ${o}_bench_${ver}.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${o}_bench_${ver}_driver_window.log 2>&1; grep '^{' ${o}_bench_${ver}_driver_window.log | tail -1 > ${o}_bench_${ver}_driver_window.json
bash tools/kstats.sh ${o}_kernel_stats_${ver}_sparse_bench.csv
bash tools/dense_pmc.sh 256 ${o}_${ver}_dense_pcg_256 > /dev/null 2>&1
bash tools/dense_pmc.sh 512 ${o}_${ver}_dense_pcg_512 > /dev/null 2>&1
# the bench line again, now that the PMC captures of THIS code state exist (bench.py reads them from profiles/)
mkdir -p profiles; for n in 256 512; do cp ${o}_${ver}_dense_pcg_${n}_pmc.json profiles/${tag}_pmc_dense_pcg_${n}.json; done
timeout 600 python bench.py > ${o}_bench_${ver}.log 2>&1; grep '^{' ${o}_bench_${ver}.log | tail -1 > ${o}_bench_${ver}.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${o}_bench_${ver}_driver_window.log 2>&1; grep '^{' ${o}_bench_${ver}_driver_window.log | tail -1 > ${o}_bench_${ver}_driver_window.json
( cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/_sq && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $root/gpurun_out/_sq -o p -- python $root/bench.py --dense-only --dense-size 256 > $root/gpurun_out/_sq.log 2>&1 )
python tools/pmc_summary.py gpurun_out/_sq > ${o}_${ver}_pmc_sq_dense_pcg_256.csv; rm -rf gpurun_out/_sq
for tr in direct host; do for n in 2 4 8; do python tools/slab_loopback_bench.py corner_dams_256 $n 60 5 single_reduction 1 $tr; done; done > ${o}_${ver}_slab_loopback.jsonl 2>${o}_slab.err
# (round 5) strong scaling of ONE domain on one GPU: uniform / weighted / dynamic cuts, per-slab GPU-busy time
for m in uniform weighted dynamic; do timeout 300 python tools/slab_cuts_bench.py corner_dams_256 8 60 10 $m coarse direct; done > ${o}_${ver}_slab_cuts.jsonl 2>>${o}_slab.err
bash tools/dense_sweep.sh 256 256:16 512:16 512:32 1024:16 1024:32 > ${o}_${ver}_dense_sweep.txt 2>&1
bash tools/dense_sweep.sh 512 512:16 512:32 1024:32 >> ${o}_${ver}_dense_sweep.txt 2>&1
bash tools/headline_pmc.sh ${o}_${ver}_pmc_headline > /dev/null 2>&1
python bench.py --transfer-only 2>/dev/null | grep '^{' | tail -1 > ${o}_${ver}_transfer_microbench_256.json
for sc in dam_halfhalf double_dam dam_halfhalf_highres corner_dams_512 corner_dams_128 single_cell_debug wavegenerator_cube; do
  python bench.py --scene $sc --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$sc', d['value'], 'ref', d['value_reference_schedule'], d['pcg_iters_per_step'])"
done > ${o}_${ver}_other_scenes.txt 2>&1
cat ${o}_${ver}_other_scenes.txt ${o}_${ver}_slab_loopback.jsonl ${o}_${ver}_dense_sweep.txt ${o}_${ver}_dense_pcg_256_pmc.txt ${o}_${ver}_dense_pcg_512_pmc.txt
python - <<P
import json
for f in ("${o}_bench_${ver}.json", "${o}_bench_${ver}_driver_window.json"):
    d=json.load(open(f))
    print(f, "value", d["value"], "ref", d["value_reference_schedule"], "ff", (d.get("fast_forward") or {}).get("steps_per_s"))
    for k in ("roofline", "roofline_512"):
        r=d.get(k)
        if r: print(" ", k, r["frac"], r["avg_us"], "KD", r["second_kernel"]["frac"], r["second_kernel"]["avg_us"], "fused", r["iteration_frac_fused_pair"], "traffic ok", r.get("traffic_capture_matches_sources"))
    print("  breakdown", d["kernel_breakdown"])
P
