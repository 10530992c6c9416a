#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
o=gpurun_out/r05_e
for m in 1 0 1 0; do
python bench.py --no-cpu-baseline --no-dense-pcg --no-other-schedule --no-fast-forward --profile-steps 0 --tune pcg_tail_margin=$m 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('margin $m', d['value'], d['pcg_iters_per_step'], d.get('dispatches_per_step'))"
done
python bench.py --no-cpu-baseline --no-dense-pcg --no-fast-forward 2>/dev/null | tail -1 > ${o}_bench.json; python -c "
import json; d=json.load(open('${o}_bench.json')); print(d['value'], d['value_reference_schedule'], d['kernel_breakdown'])"
