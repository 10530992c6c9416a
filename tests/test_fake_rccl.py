"""tests/native/fake_rccl.cpp (TEST INFRASTRUCTURE: the RCCL entry points of the z-slab transport between processes that share one GPU)
-- its host-memory build, exercised on the CPU with real processes: grouped send / receive larger than the ring buffers, all-gather,
all-reduce, and the failure model (a rank that aborts or dies makes its peers' operations fail instead of hanging).  The GPU build of
the same file carries the multi-rank slab tests of tests/test_gpu_multirank.py."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT

SRC = os.path.join(ROOT, "tests", "native", "fake_rccl.cpp")
HOST_LIB = os.path.join(ROOT, "tests", "native", "libfake_rccl_host.so")
F32, U8 = 7, 1      # ncclFloat32, ncclUint8
SUM, MAX = 0, 2     # ncclSum, ncclMax


class UniqueId(C.Structure):      # ncclUniqueId is passed BY VALUE to ncclCommInitRank
    _fields_ = [("internal", C.c_char * 128)]


def build_host_lib():
    if not os.path.exists(HOST_LIB) or os.path.getmtime(HOST_LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DFAKE_RCCL_HOST_MEMORY", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC, "-o", HOST_LIB])
    return HOST_LIB


def _lib():
    L = C.CDLL(build_host_lib())
    vp = C.c_void_p
    L.ncclGetUniqueId.argtypes = [vp]
    L.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, UniqueId, C.c_int]
    L.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    L.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.ncclCommDestroy.argtypes = [vp]
    L.ncclCommAbort.argtypes = [vp]
    return L


def _init(L, rank, world, uid):
    comm = C.c_void_p()
    buf = UniqueId.from_buffer_copy(uid)
    assert L.ncclCommInitRank(C.byref(comm), world, buf, rank) == 0
    return comm


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _worker(rank, world, uid, mode, q):
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "5"
    L = _lib()
    comm = _init(L, rank, world, uid)
    try:
        if mode == "exchange":
            # a ring of grouped exchanges, 3 MiB each way (3x the ring buffer of a channel), plus a small message behind it on the same channel
            up, dn = (rank + 1) % world, (rank - 1) % world
            big = np.full(3 << 18, float(rank), np.float32) + np.arange(3 << 18, dtype=np.float32) * 1e-6
            small = np.array([rank * 10 + 1, rank * 10 + 2], np.float32)
            got_big, got_small = np.zeros_like(big), np.zeros_like(small)
            assert L.ncclGroupStart() == 0
            assert L.ncclSend(_p(big), big.size, F32, up, comm, None) == 0
            assert L.ncclSend(_p(small), small.size, F32, up, comm, None) == 0
            assert L.ncclRecv(_p(got_big), got_big.size, F32, dn, comm, None) == 0
            assert L.ncclRecv(_p(got_small), got_small.size, F32, dn, comm, None) == 0
            assert L.ncclGroupEnd() == 0
            exp = np.full(3 << 18, float(dn), np.float32) + np.arange(3 << 18, dtype=np.float32) * 1e-6
            ok = np.array_equal(got_big, exp) and np.array_equal(got_small, np.array([dn * 10 + 1, dn * 10 + 2], np.float32))
            # all-gather and all-reduce
            seg = np.arange(5, dtype=np.float32) + 100 * rank
            allg = np.zeros(5 * world, np.float32)
            allg[5 * rank:5 * rank + 5] = seg       # in place, like the slab transport
            assert L.ncclAllGather(_p(allg[5 * rank:]), _p(allg), 5, F32, comm, None) == 0
            ok = ok and np.array_equal(allg, np.concatenate([np.arange(5, dtype=np.float32) + 100 * r for r in range(world)]))
            v = np.array([rank + 1.0, -rank], np.float32)
            out = np.zeros(2, np.float32)
            assert L.ncclAllReduce(_p(v), _p(out), 2, F32, MAX, comm, None) == 0
            ok = ok and np.array_equal(out, np.array([world, 0], np.float32))
            assert L.ncclAllReduce(_p(v), _p(out), 2, F32, SUM, comm, None) == 0
            ok = ok and np.array_equal(out, np.array([world * (world + 1) / 2, -world * (world - 1) / 2], np.float32))
            q.put((rank, "ok" if ok else "mismatch"))
        elif mode in ("abort", "die"):
            # rank 1 leaves (abort / process exit) while the others wait for its message
            if rank == 1:
                if mode == "abort":
                    L.ncclCommAbort(comm)
                    comm = None
                    q.put((rank, "aborted"))
                else:
                    q.put((rank, "dying"))
                    q.close(); q.join_thread()      # (flush the queue's feeder thread before the hard exit)
                    os._exit(0)
            else:
                x = np.zeros(4, np.float32)
                rc = L.ncclRecv(_p(x), 4, F32, 1, comm, None)
                q.put((rank, "error %d" % rc if rc != 0 else "unexpected success"))
    finally:
        if comm is not None:
            L.ncclCommDestroy(comm)


def _run(world, mode):
    L = _lib()
    uid = UniqueId()
    assert L.ncclGetUniqueId(C.byref(uid)) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, bytes(uid), mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
    return out


@pytest.mark.parametrize("world", [2, 4, 8])
def test_grouped_exchange_allgather_allreduce(world):
    out = _run(world, "exchange")
    assert out == {r: "ok" for r in range(world)}, out


@pytest.mark.parametrize("mode", ["abort", "die"])
def test_a_rank_that_leaves_makes_its_peers_fail_instead_of_hanging(mode):
    out = _run(3, mode)
    assert out[0].startswith("error") and out[2].startswith("error"), out
