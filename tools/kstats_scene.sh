#!/bin/bash
# usage (GPU box): tools/kstats_scene.sh SCENE OUT.csv [STEPS]
scene=$1; out=$2; steps=${3:-60}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/kt
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/kt -o t -- python $root/bench.py --scene $scene --steps $steps --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 --no-other-schedule > $root/gpurun_out/kt.log 2>&1
cd $root
python tools/rocprof_summary.py $(find gpurun_out/kt -name "*.db" | head -1) > $out
