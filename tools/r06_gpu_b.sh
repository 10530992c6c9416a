#!/bin/bash
# round 6: ablations of the run-form gather on the headline scene (p2g_gather_ablate: 1 = no row fetch, 2 = no walk) against the linked-list form
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r06_b
timeout 600 python -m pytest tests/test_gpu_vs_ref.py tests/test_gpu_parity.py -m gpu -q -x --maxfail=5 > ${o}_tests.log 2>&1; tail -5 ${o}_tests.log
for t in "p2g_runs=1" "p2g_gather_ablate=1" "p2g_gather_ablate=2" "p2g_gather_ablate=3" "p2g_runs=0"; do
  timeout 300 python bench.py --no-cpu-baseline --no-dense-pcg --no-other-schedule --no-fast-forward --tune $t > ${o}_bench.log 2>&1
  grep '^{' ${o}_bench.log | tail -1 > ${o}_bench.json
  python - <<P
import json
d=json.load(open("${o}_bench.json"))
u=d["kernel_breakdown"]["us_per_step"]
print("$t value", d["value"], {k: u[k] for k in ("gather_velocity","build_lists","correct","reset_bricks","advect","density_gather","brick_lists")}, "sum", d["kernel_breakdown"]["sum_us_per_step"])
P
done
