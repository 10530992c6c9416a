#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
for t in p2g_runs=1 p2g_runs=0; do
timeout 900 python bench.py --transfer-only --tune $t 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('random_order','after_binning'):
    print('$t', k, {q: (v['avg_us'], v['frac']) for q,v in d[k].items() if isinstance(v, dict)})
"
done
