#!/bin/bash
# round 6: list-centric density gather -- parity first, then M4 and the two big scenes against the tile-centric form
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_vs_ref.py -q -x -m gpu -p no:cacheprovider -k "density or every_stage or caps" -s 2>&1 | grep -v "^corner\|^dam_\|^double" | tail -12
for own in 1 0; do
  timeout 600 python bench.py --transfer-only --tune p2g_own=$own 2>&1 | grep '^{' | tail -1 > gpurun_out/r06_density_own${own}_m4.json
  python - <<P
import json
d=json.load(open("gpurun_out/r06_density_own${own}_m4.json"))
for k in ("random_order","after_binning"):
    print("M4 p2g_own=$own", k, {c: (d[k][c]["avg_us"], d[k][c]["launches"], d[k][c]["frac"]) for c in ("density_gather","gather_velocity")})
P
done
for sc in dam_halfhalf_highres corner_dams_512; do
  for own in 1 0; do
    timeout 600 python bench.py --scene $sc --steps 60 --no-cpu-baseline --no-dense-pcg --no-fast-forward --no-other-schedule --tune p2g_own=$own 2>/dev/null | grep '^{' | tail -1 > gpurun_out/_k.json
    python - <<P
import json
d=json.load(open("gpurun_out/_k.json"))
kb=d["kernel_breakdown"]["us_per_step"]
print("$sc p2g_own=$own value", d["value"], "density_gather", kb.get("density_gather"), "gather_velocity", kb.get("gather_velocity"))
P
  done
done
