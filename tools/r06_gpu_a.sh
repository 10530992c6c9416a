#!/bin/bash
# round 6, first look at the run-form P2G lists: the tests that touch the transfer, then A/B of the two list forms on the headline scene
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r06_a
timeout 900 python -m pytest tests/test_gpu_vs_ref.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_abi.py -m gpu -q -x --maxfail=5 > ${o}_tests.log 2>&1; tail -30 ${o}_tests.log
for runs in 1 0; do
  timeout 300 python bench.py --no-cpu-baseline --no-dense-pcg --no-other-schedule --no-fast-forward --tune p2g_runs=$runs > ${o}_bench_runs$runs.log 2>&1
  grep '^{' ${o}_bench_runs$runs.log | tail -1 > ${o}_bench_runs$runs.json
  python - <<P
import json
d=json.load(open("${o}_bench_runs$runs.json"))
print("p2g_runs=$runs value", d["value"], "breakdown", d["kernel_breakdown"])
P
done
