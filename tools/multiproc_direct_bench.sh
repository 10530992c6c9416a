#!/bin/bash
# usage (GPU box): tools/multiproc_direct_bench.sh N [transport] -- `bench.py --gpus N` with N processes on the ONE GPU of the box (strong scaling of
# corner_dams_256): the RCCL stand-in (tests/native/libfake_rccl.so) only carries the group's creation, the data plane is the DIRECT transport over
# hipIpc (or, transport = rccl, the host-staged stand-in itself: sequencing, not timing).  All slabs share one GPU, so the single-domain speed is the ceiling.
n=${1:-2}; tr=${2:-auto}
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
d=$(mktemp -d)
LD_PRELOAD=$root/tests/native/libfake_rccl.so FAKE_RCCL_DIR=$d BLUB_BENCH_BACKEND=gloo BLUB_BENCH_TRANSPORT=$tr HSA_ENABLE_IPC_MODE_LEGACY=0 \
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
  bench.py --gpus $n --steps 60 --warmup 10 --no-dense-pcg 2>/dev/null | grep '^{' | tail -1
rm -rf $d
