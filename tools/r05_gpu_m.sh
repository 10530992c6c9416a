#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
d=$(mktemp -d)
LD_PRELOAD=$root/tests/native/libfake_rccl.so FAKE_RCCL_DIR=$d BLUB_BENCH_BACKEND=gloo BLUB_BENCH_TRANSPORT=direct HSA_ENABLE_IPC_MODE_LEGACY=0 BLUB_BENCH_NO_SECONDARY=1 \
 BLUB_BENCH_STALL=2:9:10 BLUB_BENCH_REBALANCE_EVERY=4 BLUB_BENCH_CHECKPOINT_INTERVAL=4 \
 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29655 \
 bench.py --gpus 4 --steps 24 --warmup 4 --no-dense-pcg --scene corner_dams_128 2>gpurun_out/r05_m_err.log | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['transport'], 'recovered', d['recovered_in_place'], 'recuts', c['recuts_in_run'], c['slab_cuts'], c['slab_cuts_at_end'])"
grep -i "recovered\|timed out\|error" gpurun_out/r05_m_err.log | head -5
rm -rf $d
