#!/bin/bash
# One GPU-box session of round 3 (everything a measurement pass needs, in one gpurun call).  usage: tools/gpu_call.sh TAG [tests|notests]
tag=${1:-a}; mode=${2:-tests}
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r03_${tag}
if [ "$mode" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -s > ${o}_gpu_tests.log 2>&1; tail -25 ${o}_gpu_tests.log
fi
timeout 600 python bench.py > ${o}_bench.log 2>&1; grep '^{' ${o}_bench.log | tail -1 > ${o}_bench.json
python - <<P
import json
d=json.load(open("${o}_bench.json"))
print("value", d["value"], "ref", d["value_reference_schedule"], "single", d["value_single_reduction_schedule"], "ff", (d.get("fast_forward") or {}).get("steps_per_s"))
for k in ("roofline", "roofline_512"):
    r=d.get(k)
    if r: print(k, r["frac"], r["avg_us"], "KD", r["second_kernel"]["frac"], r["second_kernel"]["avg_us"], "fused", r["iteration_frac_fused_pair"])
print("breakdown", d["kernel_breakdown"])
print("cpu", d.get("cpu_baseline"))
P
