#!/bin/bash
# Copies the files of one tools/gpu_final.sh pass (gpurun_out/) into profiles/ and drops the versioned files of an older pass.
# usage: tools/collect_profiles.sh r03 v8 [v7]
tag=$1; ver=$2; old=${3:-}
g=gpurun_out; p=profiles
cp $g/${tag}_bench_${ver}.json $g/${tag}_bench_${ver}_driver_window.json $g/${tag}_gpu_tests_${ver}.log $g/${tag}_kernel_stats_${ver}_sparse_bench.csv $g/${tag}_smoke_${ver}.log $p/
for n in 256 512; do
  cp $g/${tag}_${ver}_dense_pcg_${n}_kernel_stats.csv $p/${tag}_kernel_stats_dense_pcg_${n}.csv
  cp $g/${tag}_${ver}_dense_pcg_${n}_pmc.json $p/${tag}_pmc_dense_pcg_${n}.json
  cp $g/${tag}_${ver}_dense_pcg_${n}_pmc.txt $p/${tag}_pmc_dense_pcg_${n}.txt
done
cp $g/${tag}_${ver}_other_scenes.txt $p/${tag}_other_scenes.txt
cp $g/${tag}_${ver}_pmc_sq_dense_pcg_256.csv $p/${tag}_pmc_sq_dense_pcg_256.csv
cp $g/${tag}_${ver}_slab_loopback.jsonl $p/${tag}_slab_loopback.jsonl
cp $g/${tag}_${ver}_transfer_microbench_256.json $p/${tag}_transfer_microbench_256.json
[ -f $g/${tag}_${ver}_dense_sweep.txt ] && cp $g/${tag}_${ver}_dense_sweep.txt $p/${tag}_dense_sweep.txt
for i in 1 2 3 4; do [ -f $g/${tag}_${ver}_pmc_headline_$i.csv ] && cp $g/${tag}_${ver}_pmc_headline_$i.csv $p/${tag}_pmc_headline_$i.csv; done
if [ -n "$old" ]; then git rm -q --cached $p/${tag}_*_${old}*.* 2>/dev/null; rm -f $p/${tag}_*_${old}*.*; fi
sed -i "s/${tag}_bench_${old}/${tag}_bench_${ver}/g; s/${tag}_kernel_stats_${old}/${tag}_kernel_stats_${ver}/g; s/${tag}_gpu_tests_${old}/${tag}_gpu_tests_${ver}/g; s/${tag}_smoke_${old}/${tag}_smoke_${ver}/g; s/\*\*${old}\*\* files (\`tools\/gpu_final.sh ${tag} ${old}\`/**${ver}** files (\`tools\/gpu_final.sh ${tag} ${ver}\`/" $p/README.md DESIGN.md README.md
