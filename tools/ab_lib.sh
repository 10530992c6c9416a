for rep in 1 2 3; do
  for lib in "" build/libblubhip_bpb4.so build/libblubhip_bpb1.so; do
    out=$(BLUBHIP_LIB=$([ -n "$lib" ] && echo $PWD/$lib) python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['pcg_iters_per_step'], 'ref', d['value_reference_schedule'])")
    echo "[${lib:-default bpb2}] $out"
  done
done
