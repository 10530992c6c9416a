#!/bin/bash
# usage (on the GPU box): tools/slab_kstats.sh OUT.csv [slabs] [transport]  -- rocprofv3 kernel trace of the z-slab loopback bench -> per-kernel table
out=${1:-gpurun_out/slab_kstats.csv}; slabs=${2:-2}; transport=${3:-direct}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/kts
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/kts -o t -- python $root/tools/slab_loopback_bench.py corner_dams_256 $slabs 30 5 single_reduction 1 $transport > $root/gpurun_out/kts.log 2>&1
cd $root
python tools/rocprof_summary.py $(find gpurun_out/kts -name "*.db" | head -1) > $out
