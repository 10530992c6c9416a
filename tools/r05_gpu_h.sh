#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r05_h
timeout 600 python -m pytest tests/test_gpu_slab_cuts.py -q -s -k "moving or rebalanc" > ${o}_tests.log 2>&1; grep -n "rebalance\|passed\|failed" ${o}_tests.log | cut -c1-300
for m in uniform dynamic; do timeout 300 python tools/slab_cuts_bench.py corner_dams_256 8 60 10 $m coarse direct; done > ${o}_slab_cuts_dynamic.jsonl 2>${o}_err.log
for m in uniform dynamic; do timeout 600 python tools/slab_cuts_bench.py corner_dams_512 8 60 10 $m coarse direct; done >> ${o}_slab_cuts_dynamic.jsonl 2>>${o}_err.log
for m in uniform dynamic; do timeout 600 python tools/slab_cuts_bench.py corner_dams_512 8 20 5 $m coarse direct; done >> ${o}_slab_cuts_dynamic.jsonl 2>>${o}_err.log
python - <<P
import json
for l in open("${o}_slab_cuts_dynamic.jsonl"):
    d=json.loads(l); print(d["scene"], d["cuts_mode"], d["steps"], "recuts", d["recuts"], "cuts_end", d["cuts_at_end"], "busiest", d["busiest_slab_us_per_step"], "sum", d["sum_of_slabs_us_per_step"], "wall", d["wall_ms_per_step_all_slabs_on_one_gpu"], "bricks_end", d["fluid_bricks_per_slab_at_end"])
P
tail -3 ${o}_err.log
