#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r05_c
for m in uniform weighted; do timeout 300 python tools/slab_cuts_bench.py corner_dams_256 8 20 5 $m coarse direct; done > ${o}_slab_cuts_driver_window.jsonl 2>${o}_cuts.err
for m in uniform weighted; do timeout 600 python tools/slab_cuts_bench.py corner_dams_512 8 20 5 $m coarse direct; timeout 600 python tools/slab_cuts_bench.py corner_dams_512 8 60 10 $m coarse direct; done > ${o}_slab_cuts_512.jsonl 2>>${o}_cuts.err
cat ${o}_slab_cuts_driver_window.jsonl ${o}_slab_cuts_512.jsonl | cut -c1-1200
BLUB_BENCH_CUTS=uniform bash tools/multiproc_direct_bench.sh 4 direct default > ${o}_multiproc.jsonl 2>&1
for hwq in default 2; do bash tools/multiproc_direct_bench.sh 8 direct $hwq; cp gpurun_out/_mp_err.log ${o}_mp8_${hwq}.err; done >> ${o}_multiproc.jsonl 2>&1
NO_SECONDARY=0 bash tools/multiproc_direct_bench.sh 2 auto auto >> ${o}_multiproc.jsonl 2>&1
python - <<P
import json
for l in open("${o}_multiproc.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print(d.get("requested"), d.get("value"), d.get("transport"), d.get("config",{}).get("slab_cuts_mode"), d.get("secondary"))
P
