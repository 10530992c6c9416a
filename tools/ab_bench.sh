#!/bin/bash
# A/B of tuning knobs on the headline window: tools/ab_bench.sh "name=value name=value" "..." -- alternates the variants, 3 runs each, prints steps/s
for rep in 1 2 3; do
  for v in "$@"; do
    args=""; for kv in $v; do [ "$kv" = "-" ] || args="$args --tune $kv"; done
    out=$(python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 $args 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['pcg_iters_per_step'], 'ref', d['value_reference_schedule'])")
    echo "[$v] $out"
  done
done
