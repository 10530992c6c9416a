#!/bin/bash
# usage: tools/dense_sweep.sh SIZE "ENV=VAL ..." ...   -- dense PCG micro-benchmark per variant (2 repetitions each)
size=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    out=$(env $v python bench.py --dense-only --dense-size $size 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('iter %.1f us frac %.4f fused %.4f | dir %.2f us %.4f | upd %.2f us %.4f' % (d['us_per_iteration_kernels'], d['iter_frac'], d['iter_frac_fused'], k['pcg_dir']['avg_us'], k['pcg_dir']['frac'], k['pcg_update']['avg_us'], k['pcg_update']['frac']))")
    echo "[$size $v] $out"
  done
done
