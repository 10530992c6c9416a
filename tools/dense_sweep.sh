#!/bin/bash
# Tile-geometry sweep of the dense 2.5-D PCG kernels on the GPU box: one line per configuration.
# usage: tools/dense_sweep.sh SIZE "T:ZC[:KUVARIANT] ..."        (T = tile width in quads, ZC = planes per tile, 0 = default)
size=${1:-256}; shift
for cfg in ${@:-0:0}; do
  IFS=: read t zc kv <<< "$cfg"; kv=${kv:-0}
  python bench.py --dense-only --dense-size $size --dense-tile-quads $t --dense-tile-planes $zc 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('dense %s T=%s zc=%s : KD %.2f us (%.3f)  KU %.2f us (%.3f)  iter %.1f us  fused frac %.3f' % (d['grid'], '$t', '$zc', k['pcg_dir']['avg_us'], k['pcg_dir']['frac'], k['pcg_update']['avg_us'], k['pcg_update']['frac'], d['us_per_iteration_kernels'], d['iter_frac_fused']))"
done
