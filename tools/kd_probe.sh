#!/bin/bash
# dense direction kernel: tile geometry x pipeline depth x store policy.  usage: tools/kd_probe.sh SIZE "T ZC [name=value ...]" ...
size=$1; shift
for cfg in "$@"; do
  set -- $cfg; t=$1; zc=$2; shift 2; extra=""; for kv in "$@"; do extra="$extra --tune $kv"; done
  python bench.py --dense-only --dense-size $size --dense-tile-quads $t --dense-tile-planes $zc $extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$size^3 $cfg : KD %.2f us (%.3f)  KU %.2f us (%.3f)  fused %.3f' % (k['pcg_dir']['avg_us'], k['pcg_dir']['frac'], k['pcg_update']['avg_us'], k['pcg_update']['frac'], d['iter_frac_fused']))"
done
