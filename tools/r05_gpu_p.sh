#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r05_p
for m in uniform dynamic; do timeout 600 python tools/slab_cuts_bench.py corner_dams_512 8 60 10 $m coarse direct; done > ${o}_slab_cuts_512.jsonl 2>${o}_err.log
timeout 300 python tools/slab_cuts_bench.py corner_dams_256 8 60 10 uniform coarse direct >> ${o}_slab_cuts_512.jsonl 2>>${o}_err.log
for tr in direct host; do for n in 2 4 8; do python tools/slab_loopback_bench.py corner_dams_256 $n 60 5 single_reduction 1 $tr; done; done > ${o}_slab_loopback.jsonl 2>>${o}_err.log
python - <<P
import json
for l in open("${o}_slab_cuts_512.jsonl"):
    d=json.loads(l); print(d["scene"], d["cuts_mode"], "busiest", d["busiest_slab_us_per_step"], "pcg", d["pcg_us_per_step_per_slab"], "wall", d["wall_ms_per_step_all_slabs_on_one_gpu"])
for l in open("${o}_slab_loopback.jsonl"):
    d=json.loads(l); print(d["slabs_on_one_gpu"], d["transport"], d["slab_group_ms_per_step"], d["protocol_overhead_vs_n_sequential_copies"])
P
NO_SECONDARY=1 bash tools/multiproc_direct_bench.sh 4 direct default | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('4 procs', d['value'], d['transport'])"
NO_SECONDARY=1 bash tools/multiproc_direct_bench.sh 2 direct default | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 procs', d['value'], d['transport'])"
