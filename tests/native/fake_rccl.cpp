// fake_rccl: the handful of RCCL entry points libblubhip's z-slab transport uses, between PROCESSES THAT SHARE ONE GPU (or none).
//
// TEST INFRASTRUCTURE, not product code.  RCCL refuses communicators with two ranks on the same device, and the development / CI box has
// one GPU -- so the multi-rank code paths of blub_slab.inc.hip (rank sequencing, grouped send / receive, partial gathers, calibration,
// abort) could never run with more than one rank (round-2 review).  This library is LD_PRELOADed in front of librccl.so: the same
// libblubhip.so, the same protocol code, N real processes.  Payloads travel through a file-backed shared mapping (host staging: device
// -> host copy, ring buffer per ordered rank pair, host -> device copy); every call blocks the host until its operations are complete,
// which is a legal (if slow) execution of RCCL's stream semantics.  Nothing here is fast and nothing here is meant to be.
//
//   build (GPU)  : hipcc -O2 -fPIC -shared tests/native/fake_rccl.cpp -o tests/native/libfake_rccl.so
//   build (host) : g++  -O2 -fPIC -shared -DFAKE_RCCL_HOST_MEMORY -I/opt/rocm/include tests/native/fake_rccl.cpp -o tests/native/libfake_rccl_host.so
//                  ("device" pointers are host pointers: the protocol itself is unit-tested on the CPU, tests/test_fake_rccl.py)
//
// Failure model: a rank that calls ncclCommAbort, or whose process disappears, makes every peer's pending / next operation return
// ncclRemoteError (after FAKE_RCCL_TIMEOUT_S seconds without progress at the latest, default 20) instead of blocking forever.
#ifndef FAKE_RCCL_HOST_MEMORY
#include <hip/hip_runtime.h>
#endif
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t RING_BYTES = 1u << 20;          // per ordered rank pair; larger messages stream through
constexpr uint32_t MAGIC = 0x46524343u;          // "FRCC"

struct Channel {                                  // single producer (src), single consumer (dst)
    std::atomic<uint64_t> head;                   // bytes written so far
    std::atomic<uint64_t> tail;                   // bytes consumed so far
    char pad[48];
    char data[RING_BYTES];
};
struct RankState { std::atomic<int> pid; std::atomic<int> aborted; std::atomic<int> left; char pad[52]; };
struct Shared {
    std::atomic<uint32_t> magic;
    std::atomic<int> joined;
    int nranks;
    RankState rank[MAX_RANKS];
    Channel ch[MAX_RANKS][MAX_RANKS];             // ch[src][dst]
};

struct Op { bool send; int peer; char* dev; size_t bytes; size_t done; bool header_done; std::vector<char> host; void* stream; };

double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
double timeout_s() { const char* e = getenv("FAKE_RCCL_TIMEOUT_S"); return e ? atof(e) : 20.0; }
bool pid_gone(int pid) {
    if (pid <= 0) return false;
    if (kill(pid, 0) != 0 && errno == ESRCH) return true;
    char path[64], buf[256];
    snprintf(path, sizeof path, "/proc/%d/stat", pid);
    FILE* f = fopen(path, "r");
    if (!f) return true;
    const size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[n] = 0;
    const char* p = strrchr(buf, ')');            // state follows the command name
    return p && (p[2] == 'Z' || p[2] == 'X');
}

int dev_to_host(void* dst, const void* src, size_t n, void* stream) {
#ifdef FAKE_RCCL_HOST_MEMORY
    (void)stream; memcpy(dst, src, n); return 0;
#else
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;      // everything enqueued before the call has produced its data
    return hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
#endif
}
int host_to_dev(void* dst, const void* src, size_t n, void* stream) {
#ifdef FAKE_RCCL_HOST_MEMORY
    (void)stream; memcpy(dst, src, n); return 0;
#else
    if (hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : 1;
#endif
}
int dev_to_dev(void* dst, const void* src, size_t n, void* stream) {
#ifdef FAKE_RCCL_HOST_MEMORY
    (void)stream; memmove(dst, src, n); return 0;
#else
    if (hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return 1;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : 1;
#endif
}

size_t dtype_size(ncclDataType_t t) {
    switch ((int)t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

}  // namespace

struct ncclComm {                                  // what ncclComm_t points to
    Shared* sh = nullptr;
    int rank = 0, nranks = 0;
    char path[160] = {0};
    int group_depth = 0;
    std::vector<Op> pending;
    bool dead = false;
};

namespace {

thread_local int g_group_depth = 0;                // NCCL groups are per thread, not per communicator
thread_local std::vector<ncclComm*> g_group_comms;

// Runs the pending operations of one communicator to completion.  Operations on one channel are served in order; sends and receives of
// different channels make progress together, so grouped exchanges of any size cannot deadlock on the ring capacity.
ncclResult_t progress(ncclComm* c) {
    if (c->dead) return ncclInvalidUsage;
    Shared* sh = c->sh;
    std::vector<Op>& ops = c->pending;
    for (Op& o : ops)                               // stage the payloads of the sends (device -> host)
        if (o.send) { o.host.resize(o.bytes); if (o.bytes && dev_to_host(o.host.data(), o.dev, o.bytes, o.stream)) return ncclUnhandledCudaError; }
        else o.host.resize(o.bytes);
    std::deque<size_t> q_send[MAX_RANKS], q_recv[MAX_RANKS];
    for (size_t i = 0; i < ops.size(); ++i) (ops[i].send ? q_send : q_recv)[ops[i].peer].push_back(i);
    size_t remaining = ops.size();
    double last_progress = now_s();
    const double limit = timeout_s();
    unsigned idle = 0;
    while (remaining) {
        bool moved = false;
        for (int p = 0; p < c->nranks; ++p) {
            if (!q_send[p].empty()) {
                Op& o = ops[q_send[p].front()];
                Channel& ch = sh->ch[c->rank][p];
                const uint64_t head = ch.head.load(std::memory_order_relaxed), tail = ch.tail.load(std::memory_order_acquire);
                size_t space = RING_BYTES - (size_t)(head - tail);
                uint64_t h = head;
                if (!o.header_done && space >= 8) {
                    const uint64_t nb = o.bytes;
                    for (int k = 0; k < 8; ++k) ch.data[(h + k) % RING_BYTES] = ((const char*)&nb)[k];
                    h += 8; space -= 8; o.header_done = true; moved = true;
                }
                if (o.header_done && o.done < o.bytes && space) {
                    const size_t n = std::min(space, o.bytes - o.done);
                    for (size_t off = 0; off < n;) {
                        const size_t at = (size_t)((h + off) % RING_BYTES), run = std::min(n - off, RING_BYTES - at);
                        memcpy(ch.data + at, o.host.data() + o.done + off, run); off += run;
                    }
                    h += n; o.done += n; moved = true;
                }
                if (h != head) ch.head.store(h, std::memory_order_release);
                if (o.header_done && o.done == o.bytes) { q_send[p].pop_front(); --remaining; }
            }
            if (!q_recv[p].empty()) {
                Op& o = ops[q_recv[p].front()];
                Channel& ch = sh->ch[p][c->rank];
                const uint64_t tail = ch.tail.load(std::memory_order_relaxed), head = ch.head.load(std::memory_order_acquire);
                size_t avail = (size_t)(head - tail);
                uint64_t t = tail;
                if (!o.header_done && avail >= 8) {
                    uint64_t nb = 0;
                    for (int k = 0; k < 8; ++k) ((char*)&nb)[k] = ch.data[(t + k) % RING_BYTES];
                    t += 8; avail -= 8; o.header_done = true; moved = true;
                    if (nb != o.bytes) {
                        fprintf(stderr, "fake_rccl: rank %d expected %zu bytes from rank %d, the matching send has %llu\n", c->rank, o.bytes, p, (unsigned long long)nb);
                        sh->rank[c->rank].aborted.store(1);
                        return ncclInvalidArgument;
                    }
                }
                if (o.header_done && o.done < o.bytes && avail) {
                    const size_t n = std::min(avail, o.bytes - o.done);
                    for (size_t off = 0; off < n;) {
                        const size_t at = (size_t)((t + off) % RING_BYTES), run = std::min(n - off, RING_BYTES - at);
                        memcpy(o.host.data() + o.done + off, ch.data + at, run); off += run;
                    }
                    t += n; o.done += n; moved = true;
                }
                if (t != tail) ch.tail.store(t, std::memory_order_release);
                if (o.header_done && o.done == o.bytes) { q_recv[p].pop_front(); --remaining; }
            }
        }
        if (moved) { last_progress = now_s(); idle = 0; continue; }
        if (++idle < 200) continue;
        idle = 0;
        for (int p = 0; p < c->nranks; ++p) {       // nothing moved for a while: is a peer we are waiting for still with us?
            if (q_send[p].empty() && q_recv[p].empty()) continue;
            if (sh->rank[p].aborted.load() || sh->rank[p].left.load() || pid_gone(sh->rank[p].pid.load())) {
                fprintf(stderr, "fake_rccl: rank %d: peer %d aborted or exited with operations pending\n", c->rank, p);
                return ncclRemoteError;
            }
        }
        if (sh->rank[c->rank].aborted.load()) return ncclInvalidUsage;
        if (now_s() - last_progress > limit) { fprintf(stderr, "fake_rccl: rank %d: no progress for %.0f s\n", c->rank, limit); return ncclRemoteError; }
        timespec ts = {0, 50000}; nanosleep(&ts, nullptr);
    }
    for (Op& o : ops)                               // deliver the received payloads (host -> device)
        if (!o.send && o.bytes && host_to_dev(o.dev, o.host.data(), o.bytes, o.stream)) return ncclUnhandledCudaError;
    ops.clear();
    return ncclSuccess;
}

ncclResult_t enqueue(ncclComm* c, bool send, int peer, const void* buf, size_t bytes, void* stream) {
    if (!c || c->dead) return ncclInvalidArgument;
    if (peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    Op o; o.send = send; o.peer = peer; o.dev = (char*)buf; o.bytes = bytes; o.done = 0; o.header_done = false; o.stream = stream;
    c->pending.push_back(std::move(o));
    if (g_group_depth == 0) { const ncclResult_t r = progress(c); if (r != ncclSuccess) c->pending.clear(); return r; }
    bool known = false;
    for (ncclComm* k : g_group_comms) known = known || k == c;
    if (!known) g_group_comms.push_back(c);
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    unsigned long long r = 0;
    FILE* f = fopen("/dev/urandom", "r");
    if (f) { if (fread(&r, sizeof r, 1, f) != 1) r = 0; fclose(f); }
    if (!r) r = (unsigned long long)(now_s() * 1e9) ^ ((unsigned long long)getpid() << 32);
    snprintf(id->internal, sizeof id->internal, "FAKERCCL-%d-%016llx", (int)getpid(), r);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (strncmp(id.internal, "FAKERCCL-", 9) != 0) return ncclInvalidArgument;
    ncclComm* c = new (std::nothrow) ncclComm();
    if (!c) return ncclSystemError;
    const char* dir = getenv("FAKE_RCCL_DIR");
    snprintf(c->path, sizeof c->path, "%s/%.100s", dir ? dir : "/tmp", id.internal);
    bool creator = false;
    int fd = open(c->path, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd >= 0) { creator = true; if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); delete c; return ncclSystemError; } }
    else {
        const double t0 = now_s();
        for (;;) {                                   // wait for the creator to size the file
            fd = open(c->path, O_RDWR);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Shared)) break;
            if (fd >= 0) close(fd);
            if (now_s() - t0 > timeout_s()) { delete c; return ncclSystemError; }
            timespec ts = {0, 1000000}; nanosleep(&ts, nullptr);
        }
    }
    void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared*)m; c->rank = rank; c->nranks = nranks;
    if (creator) { c->sh->nranks = nranks; c->sh->magic.store(MAGIC, std::memory_order_release); }   // (a fresh file is all zeros: heads, tails, flags)
    const double t0 = now_s();
    while (c->sh->magic.load(std::memory_order_acquire) != MAGIC) {
        if (now_s() - t0 > timeout_s()) { munmap(m, sizeof(Shared)); delete c; return ncclSystemError; }
        timespec ts = {0, 200000}; nanosleep(&ts, nullptr);
    }
    if (c->sh->nranks != nranks) { munmap(m, sizeof(Shared)); delete c; return ncclInvalidArgument; }
    c->sh->rank[rank].pid.store((int)getpid());
    c->sh->joined.fetch_add(1);
    while (c->sh->joined.load() < nranks) {          // rendezvous, like the real bootstrap
        if (now_s() - t0 > timeout_s()) { fprintf(stderr, "fake_rccl: rank %d: only %d of %d ranks joined\n", rank, c->sh->joined.load(), nranks); munmap(m, sizeof(Shared)); delete c; return ncclSystemError; }
        timespec ts = {0, 200000}; nanosleep(&ts, nullptr);
    }
    *comm = c;
    return ncclSuccess;
}

static void release(ncclComm* c, bool aborted) {
    if (!c || !c->sh) return;
    if (aborted) c->sh->rank[c->rank].aborted.store(1);
    c->sh->rank[c->rank].left.store(1);
    bool last = true;
    for (int r = 0; r < c->nranks; ++r) last = last && c->sh->rank[r].left.load() != 0;
    munmap(c->sh, sizeof(Shared));
    if (last) unlink(c->path);
    c->sh = nullptr; c->dead = true;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { if (!comm) return ncclInvalidArgument; release(comm, false); delete comm; return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t comm) { if (!comm) return ncclInvalidArgument; release(comm, true); delete comm; return ncclSuccess; }

ncclResult_t ncclGroupStart(void) { g_group_depth += 1; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    ncclResult_t res = ncclSuccess;
    for (ncclComm* c : g_group_comms) { const ncclResult_t r = progress(c); if (r != ncclSuccess) { c->pending.clear(); res = r; } }
    g_group_comms.clear();
    return res;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t es = dtype_size(datatype);
    if (!es) return ncclInvalidArgument;
    return enqueue(comm, true, peer, sendbuff, count * es, (void*)stream);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t es = dtype_size(datatype);
    if (!es) return ncclInvalidArgument;
    return enqueue(comm, false, peer, recvbuff, count * es, (void*)stream);
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    const size_t es = dtype_size(datatype);
    if (!comm || comm->dead || !es) return ncclInvalidArgument;
    if (g_group_depth > 0) return ncclInvalidUsage;   // (not needed by the slab transport)
    const size_t seg = sendcount * es;
    char* own = (char*)recvbuff + (size_t)comm->rank * seg;
    if (seg && own != (const char*)sendbuff && dev_to_dev(own, sendbuff, seg, (void*)stream)) return ncclUnhandledCudaError;
    ncclGroupStart();
    for (int p = 0; p < comm->nranks; ++p) {
        if (p == comm->rank) continue;
        enqueue(comm, true, p, sendbuff, seg, (void*)stream);
        enqueue(comm, false, p, (char*)recvbuff + (size_t)p * seg, seg, (void*)stream);
    }
    return ncclGroupEnd();
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!comm || comm->dead) return ncclInvalidArgument;
    if (datatype != ncclFloat32 || (op != ncclSum && op != ncclMax && op != ncclMin) || g_group_depth > 0) return ncclInvalidUsage;
    const size_t bytes = count * sizeof(float);
    std::vector<float> mine(count), all((size_t)comm->nranks * count);
    if (bytes && dev_to_host(mine.data(), sendbuff, bytes, (void*)stream)) return ncclUnhandledCudaError;
    // every rank receives every other rank's array and reduces in rank order: the same result everywhere
    std::vector<Op>& ops = comm->pending;
    for (int p = 0; p < comm->nranks; ++p) {
        if (p == comm->rank) continue;
        Op s; s.send = true; s.peer = p; s.dev = nullptr; s.bytes = bytes; s.done = 0; s.header_done = false; s.stream = nullptr;
        Op r = s; r.send = false;
        ops.push_back(std::move(s)); ops.push_back(std::move(r));
    }
    // (host-resident variant of progress(): payloads are already on the host)
    {
        Shared* sh = comm->sh;
        size_t remaining = ops.size();
        double last = now_s();
        while (remaining) {
            bool moved = false;
            for (Op& o : ops) {
                if (o.header_done && o.done == o.bytes) continue;
                Channel& ch = o.send ? sh->ch[comm->rank][o.peer] : sh->ch[o.peer][comm->rank];
                char* host = o.send ? (char*)mine.data() : (char*)(all.data() + (size_t)o.peer * count);
                uint64_t head = ch.head.load(std::memory_order_acquire), tail = ch.tail.load(std::memory_order_acquire);
                if (o.send) {
                    size_t space = RING_BYTES - (size_t)(head - tail);
                    uint64_t h = head;
                    if (!o.header_done && space >= 8) { const uint64_t nb = o.bytes; for (int k = 0; k < 8; ++k) ch.data[(h + k) % RING_BYTES] = ((const char*)&nb)[k]; h += 8; space -= 8; o.header_done = true; }
                    if (o.header_done && o.done < o.bytes && space) {
                        const size_t n = std::min(space, o.bytes - o.done);
                        for (size_t k = 0; k < n; ++k) ch.data[(h + k) % RING_BYTES] = host[o.done + k];
                        h += n; o.done += n;
                    }
                    if (h != head) { ch.head.store(h, std::memory_order_release); moved = true; }
                } else {
                    size_t avail = (size_t)(head - tail);
                    uint64_t t = tail;
                    if (!o.header_done && avail >= 8) {
                        uint64_t nb = 0; for (int k = 0; k < 8; ++k) ((char*)&nb)[k] = ch.data[(t + k) % RING_BYTES];
                        t += 8; avail -= 8; o.header_done = true;
                        if (nb != o.bytes) { ops.clear(); return ncclInvalidArgument; }
                    }
                    if (o.header_done && o.done < o.bytes && avail) {
                        const size_t n = std::min(avail, o.bytes - o.done);
                        for (size_t k = 0; k < n; ++k) host[o.done + k] = ch.data[(t + k) % RING_BYTES];
                        t += n; o.done += n;
                    }
                    if (t != tail) { ch.tail.store(t, std::memory_order_release); moved = true; }
                }
                if (o.header_done && o.done == o.bytes) --remaining;
            }
            if (moved) { last = now_s(); continue; }
            for (int p = 0; p < comm->nranks; ++p)
                if (p != comm->rank && (sh->rank[p].aborted.load() || sh->rank[p].left.load() || pid_gone(sh->rank[p].pid.load()))) { ops.clear(); return ncclRemoteError; }
            if (now_s() - last > timeout_s()) { ops.clear(); return ncclRemoteError; }
            timespec ts = {0, 50000}; nanosleep(&ts, nullptr);
        }
        ops.clear();
    }
    memcpy(all.data() + (size_t)comm->rank * count, mine.data(), bytes);
    std::vector<float> out(count);
    for (size_t i = 0; i < count; ++i) {
        float v = all[i];
        for (int p = 1; p < comm->nranks; ++p) { const float x = all[(size_t)p * count + i]; v = op == ncclSum ? v + x : (op == ncclMax ? (x > v ? x : v) : (x < v ? x : v)); }
        out[i] = v;
    }
    if (bytes && host_to_dev(recvbuff, out.data(), bytes, (void*)stream)) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
    case ncclSystemError: return "fake_rccl: system error (rendezvous file / timeout)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: invalid usage";
    case ncclRemoteError: return "fake_rccl: a peer aborted, exited or stalled";
    default: return "fake_rccl: error";
    }
}
const char* ncclGetLastError(ncclComm_t) { return ""; }
ncclResult_t ncclGetVersion(int* v) { if (v) *v = 22000; return ncclSuccess; }
const char* fake_rccl_marker(void) { return "fake_rccl"; }

}  // extern "C"
