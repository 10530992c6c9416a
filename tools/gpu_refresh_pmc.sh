#!/bin/bash
# After a source change late in a round: direct-transport tests, the two PMC captures (bench.py reads them, stamped with the source hash) and the bench lines.
tag=${1:-r04}; ver=${2:-v4}
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/${tag}
timeout 400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -m gpu -q -x -k "direct or bench_gpus_2 or second_attempt or failing or slab" > ${o}_gpu_tests_${ver}_direct.log 2>&1; tail -2 ${o}_gpu_tests_${ver}_direct.log
bash tools/dense_pmc.sh 256 ${o}_${ver}_dense_pcg_256 > /dev/null 2>&1
bash tools/dense_pmc.sh 512 ${o}_${ver}_dense_pcg_512 > /dev/null 2>&1
for n in 256 512; do cp ${o}_${ver}_dense_pcg_${n}_pmc.json profiles/${tag}_pmc_dense_pcg_${n}.json; done
timeout 600 python bench.py > ${o}_bench_${ver}.log 2>&1; grep '^{' ${o}_bench_${ver}.log | tail -1 > ${o}_bench_${ver}.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > ${o}_bench_${ver}_driver_window.log 2>&1; grep '^{' ${o}_bench_${ver}_driver_window.log | tail -1 > ${o}_bench_${ver}_driver_window.json
python - <<P
import json
for f in ("${o}_bench_${ver}.json", "${o}_bench_${ver}_driver_window.json"):
    d=json.load(open(f)); r=d["roofline"]
    print(f, "value", d["value"], "ref", d["value_reference_schedule"], "roofline", r["frac"], r["avg_us"], "traffic ok", r.get("traffic_capture_matches_sources"))
P
