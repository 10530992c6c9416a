#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r05_b
timeout 900 python -m pytest tests/test_gpu_slab_cuts.py tests/test_gpu_multirank.py -q -s -x > ${o}_tests.log 2>&1; tail -15 ${o}_tests.log
for m in uniform weighted; do for n in 4 8; do timeout 300 python tools/slab_cuts_bench.py corner_dams_256 $n 60 10 $m coarse direct; done; done > ${o}_slab_cuts.jsonl 2>${o}_cuts.err
for mem in coarse fine_grained uncached; do timeout 300 python tools/slab_cuts_bench.py corner_dams_256 2 60 10 weighted $mem direct; done > ${o}_slab_memory_modes.jsonl 2>>${o}_cuts.err
cat ${o}_slab_cuts.jsonl ${o}_slab_memory_modes.jsonl | cut -c1-900
for hwq in default 2 1; do bash tools/multiproc_direct_bench.sh 4 direct $hwq; cp gpurun_out/_mp_err.log ${o}_mp4_${hwq}.err; done > ${o}_multiproc_direct.jsonl 2>&1
bash tools/multiproc_direct_bench.sh 2 auto auto >> ${o}_multiproc_direct.jsonl 2>&1
cut -c1-700 ${o}_multiproc_direct.jsonl
