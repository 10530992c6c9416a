#!/bin/bash
# kernel trace of the driver's short window (python bench.py --steps 20 --warmup 5)
out=${1:-gpurun_out/kstats_early.csv}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/kt
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/kt -o t -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 --no-other-schedule > $root/gpurun_out/kt.log 2>&1
cd $root
python tools/rocprof_summary.py $(find gpurun_out/kt -name "*.db" | head -1) > $out
