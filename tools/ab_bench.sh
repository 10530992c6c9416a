#!/bin/bash
# usage: tools/ab_bench.sh "ENV=VAL ..." "ENV=VAL ..."   -- alternates the variants, 3 runs each, prints steps/s
for rep in 1 2 3; do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-dense-pcg --profile-steps 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['pcg_iters_per_step'])")
    echo "[$v] $out"
  done
done
