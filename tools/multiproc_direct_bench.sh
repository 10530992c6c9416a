#!/bin/bash
# usage (GPU box): tools/multiproc_direct_bench.sh N [transport] [hwq] [scene] -- `bench.py --gpus N` with N processes on the ONE GPU of the box (strong scaling of
# corner_dams_256): the RCCL stand-in (tests/native/libfake_rccl.so) only carries the group's creation, the data plane is the DIRECT transport over
# hipIpc (or, transport = rccl, the host-staged stand-in itself: sequencing, not timing).  All slabs share one GPU, so the single-domain speed is the ceiling.
# hwq: GPU_MAX_HW_QUEUES for every rank (round-4 review, item 1c: four processes x the default four hardware queues oversubscribe the device's queue slots;
# kernels that spin on words another process' kernels write then only progress when the scheduler's timer rotates the queues) -- a number, "default"
# (the runtime's own four: set explicitly so that bench.py leaves it alone) or "auto" (bench.py: shared_device_queue_limit).
n=${1:-2}; tr=${2:-auto}; hwq=${3:-auto}; scene=${4:-corner_dams_256}
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
d=$(mktemp -d)
if [ "$hwq" = "default" ]; then export GPU_MAX_HW_QUEUES=4; elif [ "$hwq" != "auto" ]; then export GPU_MAX_HW_QUEUES=$hwq; fi
LD_PRELOAD=$root/tests/native/libfake_rccl.so FAKE_RCCL_DIR=$d BLUB_BENCH_BACKEND=gloo BLUB_BENCH_TRANSPORT=$tr HSA_ENABLE_IPC_MODE_LEGACY=0 BLUB_BENCH_NO_SECONDARY=${NO_SECONDARY:-1} \
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
  bench.py --gpus $n --steps 60 --warmup 10 --no-dense-pcg --scene $scene 2>gpurun_out/_mp_err.log | grep '^{' | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
if not l: print(json.dumps({'n':$n,'transport':'$tr','hwq':'$hwq','result':'no line'})); sys.exit()
d=json.loads(l); d['requested']={'n':$n,'transport':'$tr','hwq':'$hwq'}; print(json.dumps(d))"
rm -rf $d
