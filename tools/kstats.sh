#!/bin/bash
# usage (on the GPU box): tools/kstats.sh OUT.csv  -- rocprofv3 kernel trace of the headline bench run -> per-kernel table (tools/rocprof_summary.py)
out=${1:-gpurun_out/kstats.csv}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/kt
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/kt -o t -- python $root/bench.py --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 --no-other-schedule ${KSTATS_ARGS:-} > $root/gpurun_out/kt.log 2>&1
cd $root
python tools/rocprof_summary.py $(find gpurun_out/kt -name "*.db" | head -1) > $out
