#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-dense-pcg --no-other-schedule --no-fast-forward --profile-steps 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['pcg_iters_per_step'])"; }
for rep in 1 2; do
run
run --tune bricks_two_kernel_build=1
run --tune pcg_tail=0
run --tune bricks_two_kernel_build=1 --tune pcg_tail=0
done
