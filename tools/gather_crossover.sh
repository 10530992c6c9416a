#!/bin/bash
# usage (GPU box): tools/gather_crossover.sh -- two kernel traces of the headline window with the gather form forced, the engine's own choice for comparison
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 -1; do
  rm -rf $root/gpurun_out/kt_$mode
  rocprofv3 --kernel-trace -d $root/gpurun_out/kt_$mode -o t -- python $root/bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 --no-other-schedule --tune p2g_compact=$mode > $root/gpurun_out/kt_$mode.log 2>&1
done
cd $root
db() { find gpurun_out/kt_$1 -name "*.db" | head -1; }
python tools/gather_crossover.py $(db 0) $(db 1) > gpurun_out/r06_gather_crossover.json
python tools/gather_crossover.py $(db -1) $(db 1) > gpurun_out/r06_gather_crossover_auto.json
python - <<P
import json
d=json.load(open("gpurun_out/r06_gather_crossover.json")); e=json.load(open("gpurun_out/r06_gather_crossover_auto.json"))
print("mean us per step:", d["mean_us"], "engine's own choice:", e["mean_us"]["list_centric"])
print("list-centric wins in steps", d["list_centric_wins_in_steps"])
for i in range(0, d["steps"], 10): print(i, d["list_centric_us"][i], d["compacting_us"][i], e["list_centric_us"][i])
P
rm -rf gpurun_out/kt_0 gpurun_out/kt_1 gpurun_out/kt_-1
