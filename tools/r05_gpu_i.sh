#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r05_i
BLUB_BENCH_REBALANCE_EVERY=8 bash tools/multiproc_direct_bench.sh 4 direct default > ${o}_multiproc.jsonl 2>&1; cp gpurun_out/_mp_err.log ${o}_mp4.err
BLUB_BENCH_REBALANCE_EVERY=8 NO_SECONDARY=0 bash tools/multiproc_direct_bench.sh 2 auto auto >> ${o}_multiproc.jsonl 2>&1; cp gpurun_out/_mp_err.log ${o}_mp2.err
BLUB_BENCH_REBALANCE_EVERY=8 bash tools/multiproc_direct_bench.sh 4 rccl 1 corner_dams_128 >> ${o}_multiproc.jsonl 2>&1
python - <<P
import json
for l in open("${o}_multiproc.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); c=d.get("config",{}); print(d.get("requested"), d.get("value"), d.get("transport"), c.get("slab_cuts_mode"), c.get("slab_cuts"), c.get("slab_cuts_at_end"), c.get("recuts_in_run"), d.get("recovered_in_place"), "SECONDARY", d.get("secondary"))
P
tail -5 ${o}_mp4.err
timeout 900 python -m pytest tests/test_gpu_multirank.py -q -x -k "bench" > ${o}_tests.log 2>&1; tail -5 ${o}_tests.log
