#!/bin/bash
# Placement study of the grid volumes (blub_fluid_desc::volume_shift_kib; vol_alloc in blub_fluid.hip): dense PCG kernels per shift.
# usage: tools/skew_probe.sh SIZE SHIFT_KIB...      (-1 = one hipMalloc per volume)
size=${1:-512}; shift
for k in "$@"; do
python bench.py --dense-only --dense-size $size --volume-shift-kib $k 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$size^3 volume shift %5s KiB: KD %.1f us  KU %.1f us' % ('$k', k['pcg_dir']['avg_us'], k['pcg_update']['avg_us']))"
done
