#!/bin/bash
# PMC passes over the headline bench run, per-kernel sums (tools/pmc_summary.py).  usage (GPU box): tools/headline_pmc.sh OUTPREFIX [bench args]
out=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1)); rm -rf $root/gpurun_out/_hp
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/_hp -o p -- python $root/bench.py --no-cpu-baseline --no-dense-pcg --no-fast-forward --profile-steps 0 --no-other-schedule "$@" > $root/gpurun_out/_hp.log 2>&1
  python $root/tools/pmc_summary.py $root/gpurun_out/_hp > ${root}/${out}_$i.csv
done
rm -rf $root/gpurun_out/_hp
