#!/bin/bash
# Dense PCG micro-benchmark under rocprofv3: kernel trace + FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md,
# HBM section).  usage: tools/dense_pmc.sh SIZE OUTPREFIX      (writes OUTPREFIX_kernel_stats.csv, OUTPREFIX_pmc.json / .txt)
set -u
size=$1; out=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=gpurun_out/_pmc_$size
rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats -d $d/trace -o t -- python bench.py --dense-only --dense-size $size > $d/trace.log 2>&1
python tools/rocprof_summary.py $(ls $d/trace/*.db | head -1) ${out}_kernel_stats.csv k_pcg_init_z > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d/$c -o p -- python bench.py --dense-only --dense-size $size > $d/$c.log 2>&1
done
python tools/dense_pmc_report.py $size $d ${out}_kernel_stats.csv ${out}_pmc
rm -rf $d
