#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r06_m
timeout 600 python -m pytest tests/test_gpu_vs_ref.py tests/test_gpu_parity.py -m gpu -q --maxfail=8 > ${o}_tests.log 2>&1; tail -6 ${o}_tests.log
timeout 300 python tools/p2g_probe.py --steps 45 --tunes "p2g_compact=1;p2g_compact=0,p2g_own=1;p2g_compact=0,p2g_own=0" | tail -1
timeout 300 python bench.py --scene dam_halfhalf_highres --steps 60 --no-cpu-baseline --no-dense-pcg --no-other-schedule --no-fast-forward > ${o}_bench.log 2>&1
grep '^{' ${o}_bench.log | tail -1 > ${o}_bench.json
python - <<P
import json
d=json.load(open("${o}_bench.json"))
u=d["kernel_breakdown"]["us_per_step"]
print("highres value", d["value"], {k: u.get(k) for k in ("gather_velocity","build_lists","correct","advect","density_gather")}, "sum", d["kernel_breakdown"]["sum_us_per_step"])
P
timeout 900 python bench.py --transfer-only 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('random_order','after_binning'):
    print(k, {q: (v['avg_us'], v['launches']) for q,v in d[k].items() if isinstance(v, dict)})
"
