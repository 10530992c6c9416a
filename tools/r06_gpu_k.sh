#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
for lib in "" po4 po8; do
if [ -n "$lib" ]; then export BLUBHIP_LIB=$root/blub_amd/libblubhip_$lib.so; else unset BLUBHIP_LIB; fi
timeout 900 python bench.py --transfer-only 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('after_binning',):
    print('lib=$lib', k, {q: (v['avg_us'], v['launches']) for q,v in d[k].items() if isinstance(v, dict)})
"
done
