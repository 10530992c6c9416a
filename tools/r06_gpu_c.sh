#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
for lib in "" r1 r2 r4; do
  for st in 2 45; do
    if [ -n "$lib" ]; then export BLUBHIP_LIB=$root/blub_amd/libblubhip_$lib.so; else unset BLUBHIP_LIB; fi
    timeout 200 python tools/p2g_probe.py --steps $st --tunes "p2g_runs=1;p2g_gather_ablate=1;p2g_gather_ablate=2;p2g_gather_ablate=3;p2g_runs=0" 2>&1 | tail -1
  done
done
