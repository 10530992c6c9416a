#!/bin/bash
# Regenerates the measured artefacts of a round on the GPU box into gpurun_out/<tag>_* (copy the ones to keep into profiles/).
# usage: tools/refresh_profiles.sh r02 v2
tag=${1:-r02}; ver=${2:-v2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
o=gpurun_out/${tag}
python bench.py > ${o}_bench_${ver}.log 2>&1; grep '^{' ${o}_bench_${ver}.log | tail -1 > ${o}_bench_${ver}.json
bash tools/kstats.sh ${o}_kernel_stats_${ver}_sparse_bench.csv
bash tools/dense_pmc.sh 256 ${o}_${ver}_dense_pcg_256
bash tools/dense_pmc.sh 512 ${o}_${ver}_dense_pcg_512
( cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/_sq && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $root/gpurun_out/_sq -o p -- python $root/bench.py --dense-only --dense-size 256 > $root/gpurun_out/_sq.log 2>&1 )
python tools/pmc_summary.py gpurun_out/_sq > ${o}_${ver}_pmc_sq_dense_pcg_256.csv; rm -rf gpurun_out/_sq
python tools/noop_probe.py > ${o}_${ver}_noop_probe.txt 2>&1
for n in 2 4; do python tools/slab_loopback_bench.py corner_dams_256 $n; done > ${o}_${ver}_slab_loopback.jsonl 2>${o}_slab.err
python bench.py --transfer-only 2>/dev/null | grep '^{' | tail -1 > ${o}_${ver}_transfer_microbench_256.json
python -m pytest tests -m gpu -q > ${o}_${ver}_gpu_tests.log 2>&1
tail -3 ${o}_${ver}_gpu_tests.log
