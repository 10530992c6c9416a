#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r06_h
timeout 1500 python -m pytest tests/test_gpu_vs_ref.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_abi.py tests/test_gpu_baseline_parity.py -m gpu -q -x --maxfail=5 > ${o}_tests.log 2>&1; tail -4 ${o}_tests.log
for t in "resort_every=8" ${R06_TUNES}; do
  timeout 300 python bench.py --no-cpu-baseline --no-dense-pcg --no-other-schedule --no-fast-forward --tune $t > ${o}_bench.log 2>&1
  grep '^{' ${o}_bench.log | tail -1 > ${o}_bench.json
  python - <<P
import json
d=json.load(open("${o}_bench.json"))
u=d["kernel_breakdown"]["us_per_step"]
print("$t value", d["value"], u, "sum", d["kernel_breakdown"]["sum_us_per_step"], "launches", d["kernel_breakdown"]["launches_per_step"])
P
done
