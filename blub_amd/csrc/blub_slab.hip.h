// z-slab domain decomposition -- device helpers.  The protocol itself lives in blub_slab.hip.
//
// Every slab keeps volumes in GLOBAL grid coordinates (so no kernel needs an index translation and the domain-boundary
// logic -- SOLID shell, position clamps -- stays untouched) but only WORKS on the bricks of its own z-range
// [z0, z1) (multiples of the brick depth).  What a slab needs from beyond its range arrives as
//   * ghost particles  : copies of the neighbours' particles within GHOST_MARGIN cells of the interface (P2G, density
//                        gather and the marker need them), appended behind the own particles,
//   * halo planes      : one z-plane of a volume from each neighbour after every stage that produces it,
//   * partials         : the per-block partial sums / maxima of the PCG dot products, gathered from all slabs,
//   * migrating particles after advection and after the density correction.
#pragma once
#include "blub_pcg.hip.h"

namespace blubk {

constexpr float GHOST_MARGIN = 2.0f;   // cells; covers the 2-cell reach of the P2G stencil and of the marker logic

struct SlabCounts { uint32_t n_stay, n_up, n_down, n_leave, n_holes, n_fill, pad0 /* leavers held back */, pad1; };   // zeroed before every exchange

// One atomic per wave and destination instead of one per particle: every particle of a slab passes through these kernels several
// times per step and nearly all of them go to the SAME destination (measured: 192 us for 1 M particles with per-particle atomics
// on one counter).  All 64 lanes must call; returns the slot of lanes with `pred`.
__device__ __forceinline__ uint32_t wave_alloc(uint32_t* __restrict__ counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return 0u;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// Copies (does not remove) the own particles within [zlo, zhi) into a send buffer; rows == nullptr => positions only.
// One atomic per BLOCK: the particles are binned (z slowest), so the selected ones sit in a few thousand consecutive waves that
// would otherwise all hit the one counter.
__global__ __launch_bounds__(256) void k_slab_select(uint32_t n, const float4* __restrict__ pos, const float4* __restrict__ vx, const float4* __restrict__ vy,
                                                     const float4* __restrict__ vz, float zlo, float zhi, uint32_t capacity, uint32_t* __restrict__ counter,
                                                     float4* __restrict__ out_pos, float4* __restrict__ out_vx, float4* __restrict__ out_vy, float4* __restrict__ out_vz,
                                                     const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t wcount[4], wbase;
    n = particle_count(n, n_dev, 1u);
    if (blockIdx.x * 256u >= n) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) p = pos[i];
    const bool sel = live && p.z >= zlo && p.z < zhi;
    const unsigned long long m = __ballot(sel);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wcount[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = wcount[0] + wcount[1] + wcount[2] + wcount[3]; wbase = tot ? atomicAdd(counter, tot) : 0u; }
    __syncthreads();
    if (!sel) return;
    uint32_t k = wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) k += wcount[w];
    if (k >= capacity) return;
    out_pos[k] = p;
    if (out_vx) { out_vx[k] = vx[i]; out_vy[k] = vy[i]; out_vz[k] = vz[i]; }
}

// Migration, in place.  Per step a few hundred of a slab's ~1 M particles leave it; compacting ALL of them into a second set of
// arrays (128 MB of traffic, and one counter every wave bumps: 188 us measured) is replaced by
//   mark  : particles outside [z0, z1) are copied to the up / down send buffers and their indices appended to a leave list (only
//           waves that hold a leaver touch a counter),
//   match : with L leavers the slab keeps n' = n - L particles: leavers below n' are HOLES, stayers at or above n' are FILLERS
//           (equally many by counting),
//   fill  : filler k moves into hole k.
// The order of the stayers changes slightly (tail particles move forward); the reference's own binning order is "sloppy" too.
__global__ __launch_bounds__(256) void k_slab_migrate_mark(uint32_t n, const float4* __restrict__ pos, const float4* __restrict__ vx, const float4* __restrict__ vy,
                                                           const float4* __restrict__ vz, float z0, float z1, uint32_t capacity, SlabCounts* __restrict__ counts,
                                                           float4* __restrict__ up_pos, float4* __restrict__ up_vx, float4* __restrict__ up_vy, float4* __restrict__ up_vz,
                                                           float4* __restrict__ dn_pos, float4* __restrict__ dn_vx, float4* __restrict__ dn_vy, float4* __restrict__ dn_vz,
                                                           uint32_t* __restrict__ leave_idx, const uint32_t* __restrict__ n_dev, uint32_t cap_up, uint32_t cap_dn,
                                                           float4* __restrict__ pos_rw) {
    // cap_up / cap_dn: what the MESSAGE of this exchange can carry (<= capacity, the size of the send buffers).  A leaver beyond it is
    // HELD BACK: it stays an own particle of this slab for one more exchange, its z clamped just inside the range (round-3 ADVICE: a
    // front reaching an interface that carried nothing last step used to lose what did not fit).  n_up / n_down keep counting what WANTED
    // to travel, so the next exchange of this kind sizes its message for it.
    n = particle_count(n, n_dev, 1u);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) p = pos[i];
    bool up = live && p.z >= z1, down = live && p.z < z0;
    if (__ballot(up || down) == 0ull) return;          // the usual case: nobody in this wave leaves
    const uint32_t ku = wave_alloc(&counts->n_up, up), kd = wave_alloc(&counts->n_down, down);
    if (up && ku >= cap_up) { up = false; pos_rw[i].z = nextafterf(z1, 0.0f); atomicAdd(&counts->pad0, 1u); }
    if (down && kd >= cap_dn) { down = false; pos_rw[i].z = z0; atomicAdd(&counts->pad0, 1u); }
    const uint32_t kl = wave_alloc(&counts->n_leave, up || down);
    if (up) { up_pos[ku] = p; up_vx[ku] = vx[i]; up_vy[ku] = vy[i]; up_vz[ku] = vz[i]; }
    else if (down) { dn_pos[kd] = p; dn_vx[kd] = vx[i]; dn_vy[kd] = vy[i]; dn_vz[kd] = vz[i]; }
    if (up || down) leave_idx[kl] = i;
}
__global__ __launch_bounds__(256) void k_slab_migrate_match(uint32_t n, const float4* __restrict__ pos, float z0, float z1, SlabCounts* __restrict__ counts,
                                                            const uint32_t* __restrict__ leave_idx, uint32_t* __restrict__ hole_idx, uint32_t* __restrict__ fill_idx,
                                                            const uint32_t* __restrict__ n_dev) {
    n = particle_count(n, n_dev, 1u);
    const uint32_t L = counts->n_leave;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256u >= L && !(blockIdx.x == 0)) return;      // whole block beyond the leave list (block 0 still publishes n_stay)
    const uint32_t keep = n - L;
    if (t == 0) counts->n_stay = keep;
    const bool live = t < L;
    uint32_t idx = 0u; bool hole = false, filler = false;
    if (live) {
        idx = leave_idx[t];
        hole = idx < keep;
        const float4 q = pos[keep + t];
        filler = q.z >= z0 && q.z < z1;
    }
    const uint32_t kh = wave_alloc(&counts->n_holes, hole), kf = wave_alloc(&counts->n_fill, filler);
    if (hole) hole_idx[kh] = idx;
    if (filler) fill_idx[kf] = keep + t;
}
__global__ __launch_bounds__(256) void k_slab_migrate_fill(const SlabCounts* __restrict__ counts, const uint32_t* __restrict__ hole_idx, const uint32_t* __restrict__ fill_idx,
                                                           float4* __restrict__ pos, float4* __restrict__ vx, float4* __restrict__ vy, float4* __restrict__ vz) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= counts->n_holes) return;
    const uint32_t dst = hole_idx[k], src = fill_idx[k];
    pos[dst] = pos[src]; vx[dst] = vx[src]; vy[dst] = vy[src]; vz[dst] = vz[src];
}

// Ghost particles for the density projection: mark their cells FLUID and hang them into the density linked list
// (what advect_particles.comp:176-181 does for the own particles).
__global__ __launch_bounds__(256) void k_slab_insert_density_ghosts(Grid g, uint32_t first, uint32_t count, float4* __restrict__ pos, int8_t* __restrict__ marker,
                                                                    uint32_t* __restrict__ heads, const uint32_t* __restrict__ n_dev) {
    count = particle_count(count, n_dev, 3u);
    if (blockIdx.x * 256u >= count) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const bool live = k < count;            // no early return: the wave-level list insertion needs every lane
    const uint32_t i = first + k;
    float4 p = make_float4(-8.f, -8.f, -8.f, 0.f);
    if (live) {
        p = pos[i];
        const int x = (int)p.x, y = (int)p.y, z = (int)p.z;
        if (inb(g, x, y, z)) { const int c = cidx(g, x, y, z); if (marker[c] != CELL_SOLID) marker[c] = CELL_FLUID; }
    }
    const int dx = (int)(p.x - 0.5f), dy = (int)(p.y - 0.5f), dz = (int)(p.z - 0.5f);
    const uint32_t nxt = wave_list_insert(heads, (live && inb(g, dx, dy, dz)) ? cidx(g, dx, dy, dz) : -1, i);
    if (live) reinterpret_cast<uint32_t*>(pos)[4 * (size_t)i + 3] = nxt;
}

// ---- particle exchange without the host (round 3) ---------------------------------------------------------------------------------
// Message sizes are host arguments of the transport, particle counts are only known on the device.  So a message has a FIXED capacity the
// two ends derive from the number that travelled over the same link in the same exchange of the PREVIOUS step (both know it: the sender
// from its own counters, the receiver from the header it got; it reaches the host through a pinned, sequence-tagged record, never through
// a stream synchronisation) and carries the actual count in a 16-byte header in front of the position payload.  The receiver appends what
// arrived behind its own particles with a kernel that reads both counts on the device.  The capacity is 1.5 x the previous count + 2048;
// what does not fit is HELD BACK at the sender for one exchange (migration) or left out for one step (ghost copies) and counted
// (blub_slab_group_held_back); only the particle capacity of a slab itself is a hard limit (error at the next exchange).
struct SlabXferRecord { uint32_t seq, n_up, n_down, from_below, from_above, overflow, n_own, n_ghost; };   // pinned host ring entry
// after the select / migrate kernels: headers of the two outgoing messages, the new own count of a migration, overflow of the send buffers
__global__ void k_slab_finish_send(const SlabCounts* __restrict__ counts, float4* __restrict__ hdr_up, float4* __restrict__ hdr_dn, uint32_t cap_up, uint32_t cap_dn,
                                   uint32_t* __restrict__ n_dev, int migrate) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // the header carries what actually travels; what did not fit was held back (migration) or is left out for this step (ghost copies:
    // the receiver's gathers next to the interface then miss contributions for ONE step) -- counted in n_dev[3], never an error
    const uint32_t nu = counts->n_up, nd = counts->n_down;
    // header = {what travels, what wanted to}: both ends size the NEXT message of this link from the second word, so they agree
    if (hdr_up) *hdr_up = make_float4(__uint_as_float(min(nu, cap_up)), __uint_as_float(nu), 0.f, 0.f);
    if (hdr_dn) *hdr_dn = make_float4(__uint_as_float(min(nd, cap_dn)), __uint_as_float(nd), 0.f, 0.f);
    if (nu > cap_up) n_dev[3] += nu - cap_up;
    if (nd > cap_dn) n_dev[3] += nd - cap_dn;
    if (migrate) { n_dev[0] = counts->n_stay; n_dev[1] = 0u; }
}
// after the transport: append the arrivals (headers + payloads in the staging buffers) behind the own particles, publish counts + record
struct SlabAppendArgs { const float4* below[4]; const float4* above[4]; float4* dst[4]; };    // [0] = positions (header in front), [1..3] = velocity rows
__global__ __launch_bounds__(256) void k_slab_append(SlabAppendArgs a, int narr, uint32_t cap_below, uint32_t cap_above, uint32_t capacity, uint32_t* __restrict__ n_dev, int migrate,
                                                     const SlabCounts* __restrict__ counts, SlabXferRecord* __restrict__ record, uint32_t seq, uint32_t* __restrict__ done_blocks,
                                                     const uint32_t* __restrict__ dir_error) {
    uint32_t cb = a.below[0] ? __float_as_uint(a.below[0][0].x) : 0u, ca = a.above[0] ? __float_as_uint(a.above[0][0].x) : 0u;
    const uint32_t want_b = a.below[0] ? __float_as_uint(a.below[0][0].y) : 0u, want_a = a.above[0] ? __float_as_uint(a.above[0][0].y) : 0u;
    const uint32_t own = n_dev[0];
    bool over = cb > cap_below || ca > cap_above || (uint64_t)own + cb + ca > capacity;
    if (over) { cb = min(cb, cap_below); ca = min(ca, cap_above); if ((uint64_t)own + cb + ca > capacity) { cb = 0; ca = 0; } }
    // (grid-stride: the launch grid is sized from message capacities -- with the direct transport the full particle capacity -- not from what arrived)
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < cb + ca; k += gridDim.x * 256) {
        if (k < cb) {
            for (int q = 0; q < narr; ++q) a.dst[q][own + k] = a.below[q][k + (q == 0 ? 1u : 0u)];
        } else {
            for (int q = 0; q < narr; ++q) a.dst[q][own + k] = a.above[q][k - cb + (q == 0 ? 1u : 0u)];
        }
    }
    // the last block to finish publishes the counts (every append of this launch is done by then for the kernels that follow on the stream;
    // the host only reads the record)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t t = atomicAdd(done_blocks, 1u);
        if (t == gridDim.x - 1) {
            *done_blocks = 0u;
            if (over) n_dev[2] = 1u;
            if (migrate) { n_dev[0] = own + cb + ca; n_dev[1] = 0u; } else n_dev[1] = cb + ca;
            record->n_up = counts->n_up; record->n_down = counts->n_down; record->from_below = want_b; record->from_above = want_a;
            // bit 0: the particle capacity of the slab was exceeded; bit 1 (direct transport): a bounded wait for a peer ran out in a kernel enqueued before
            // this one -- the host finds both at the start of the next exchange of this kind, without a stream synchronisation (round-4 ADVICE)
            record->overflow = (n_dev[2] ? 1u : 0u) | ((dir_error && __hip_atomic_load(dir_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 2u : 0u);
            record->n_own = n_dev[0]; record->n_ghost = n_dev[1];
            __threadfence_system();
            record->seq = seq;
        }
    }
}
// the gathered fluid-brick counts of a step (PCG grid of the NEXT step's slab solves) to the host, tagged
__global__ void k_slab_publish_cnt(const float* __restrict__ gat_cnt, int nranks, float* __restrict__ host_values, uint32_t* __restrict__ host_seq, uint32_t seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < nranks; ++k) host_values[k] = gat_cnt[k];
    __threadfence_system();
    *host_seq = seq;
}

// loopback transport: all plane copies of one halo exchange in ONE launch (a hipMemcpyAsync per plane costs ~4 us of queue time each,
// ~100 of them per step).  DIRECT transport (peer-mapped memory): the same launch with write-through stores, and the last workgroup to
// finish raises the destination slabs' flag words (see SlabDirect, blub_kernels.hip.h).
// `unit`: 0 = src, dst 16-byte aligned and bytes % 16 == 0 (every plane of a grid with nx * ny % 16 == 0, every particle message); 1 = all three 4-byte
// aligned (a 4-byte count segment; f32 planes of nx * ny % 4 != 0 cannot occur, nx % 4 == 0 is enforced); 2 = bytes (the 1-byte descriptor plane of grids
// such as 20 x 30).  Round-4 ADVICE: the unaligned case used to leave the batch (a flush that raised the exchange's flags early + a plain hipMemcpyAsync
// into the peer's memory without the acknowledgement handshake); now every store of an exchange goes through the one push kernel.
struct SlabCopy { const void* src; void* dst; uint32_t bytes; uint32_t unit; };
constexpr int SLAB_COPY_MAX = 40, SLAB_FLAG_MAX = 64;
struct SlabCopyList { SlabCopy c[SLAB_COPY_MAX]; int n; };
// Acknowledgements (processes on different GPUs only): before a push kernel of exchange `seq` writes into a peer's ghost planes / staging buffers,
// that peer must have finished every kernel that reads what the PREVIOUS exchanges delivered there.  Its own push kernel of exchange `seq` is
// enqueued behind those kernels, so it raises an acknowledgement word in each of its destinations first thing, and every push kernel waits for
// the words of the ranks it is about to write to (the exchanges are symmetric: whoever I push to pushes to me).  Not needed when all slabs
// share one stream.  (K(i) pushes without it: its targets are double buffered by iteration parity.)
struct SlabFlagList { uint32_t* f[SLAB_FLAG_MAX]; int n; uint32_t seq; uint32_t* blocks_done; uint32_t* ack_out[8]; int ack_src[8]; int n_ack; const uint32_t* ack_in; uint32_t* error; };
__device__ __forceinline__ void slab_ack_handshake(const SlabFlagList& F) {
    if (F.n_ack == 0) return;      // (uniform)
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0 && blockIdx.y == 0) for (int q = 0; q < F.n_ack; ++q) st_sys_u32(F.ack_out[q], F.seq);
        for (int q = 0; q < F.n_ack; ++q) {
            unsigned spins = 0;
            while ((int32_t)(ld_sys_u32(F.ack_in + F.ack_src[q]) - F.seq) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SLAB_SPIN_LIMIT || ((spins & 1023u) == 0u && slab_gave_up(F.error))) { if (F.error) atomicOr(F.error, 1u); break; }      // a missing peer must not hang the GPU
            }
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_slab_copy_planes(SlabCopyList L) {
    const SlabCopy c = L.c[blockIdx.y];
    if (c.unit == 0u) {
        const uint32_t n16 = c.bytes >> 4;
        const uint4* __restrict__ s = reinterpret_cast<const uint4*>(c.src);
        uint4* __restrict__ d = reinterpret_cast<uint4*>(c.dst);
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) d[i] = s[i];
    } else if (c.unit == 1u) {
        const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(c.src);
        uint32_t* __restrict__ d = reinterpret_cast<uint32_t*>(c.dst);
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < (c.bytes >> 2); i += gridDim.x * 256) d[i] = s[i];
    } else {
        const uint8_t* __restrict__ s = reinterpret_cast<const uint8_t*>(c.src);
        uint8_t* __restrict__ d = reinterpret_cast<uint8_t*>(c.dst);
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < c.bytes; i += gridDim.x * 256) d[i] = s[i];
    }
}
__global__ __launch_bounds__(256) void k_slab_push_planes(SlabCopyList L, SlabFlagList F) {
    slab_ack_handshake(F);
    if ((int)blockIdx.y < L.n) {
        const SlabCopy c = L.c[blockIdx.y];
        if (c.unit == 0u) {
            const uint32_t n16 = c.bytes >> 4;
            const float4* __restrict__ s = reinterpret_cast<const float4*>(c.src);
            float4* d = reinterpret_cast<float4*>(c.dst);
            for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) st_sys_f4(d + i, s[i]);
        } else if (c.unit == 1u) {
            const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(c.src);
            uint32_t* d = reinterpret_cast<uint32_t*>(c.dst);
            for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < (c.bytes >> 2); i += gridDim.x * 256) st_sys_u32(d + i, s[i]);
        } else {
            const uint8_t* __restrict__ s = reinterpret_cast<const uint8_t*>(c.src);
            uint8_t* d = reinterpret_cast<uint8_t*>(c.dst);
            for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < c.bytes; i += gridDim.x * 256) st_sys_u8(d + i, (uint32_t)s[i]);
        }
    }
    wait_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(F.blocks_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1u == gridDim.x * gridDim.y) {
            __hip_atomic_store(F.blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int k = 0; k < F.n; ++k) st_sys_u32(F.f[k], F.seq);
        }
    }
}
// a stream waits for the flags of an exchange (kernels enqueued behind this one find the pushed data in place)
// (skip_if_done: the flags of a PCG iteration that a finished solve never launched must not be waited for)
__global__ void k_slab_wait(const uint32_t* __restrict__ flags_in, uint32_t mask, uint32_t seq, uint32_t* __restrict__ error, const PcgCtrl* __restrict__ skip_if_done) {
    if (skip_if_done && skip_if_done->done) return;
    slab_wait_flags(flags_in, mask, seq, error);
}

// DIRECT transport of a particle exchange: the sender copies header + what travels (the count is only known on the device) straight into the
// receiver's staging buffer, which has the full particle capacity -- no message size to agree on, nothing to hold back
struct SlabParticlePush { const float4* src[4]; float4* dst[4]; };      // [0]: position message (header in front), [1..3]: velocity rows; dst nullptr: no neighbour
__global__ __launch_bounds__(256) void k_slab_push_particles(SlabParticlePush up, SlabParticlePush dn, int narr, SlabFlagList F) {
    slab_ack_handshake(F);
    const SlabParticlePush& P = blockIdx.y == 0 ? up : dn;
    if (P.dst[0]) {
        const uint32_t n = __float_as_uint(P.src[0][0].x);      // header written by k_slab_finish_send
        for (int q = 0; q < narr; ++q) {
            const uint32_t cnt = n + (q == 0 ? 1u : 0u);
            for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) st_sys_f4(P.dst[q] + i, P.src[q][i]);
        }
    }
    wait_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(F.blocks_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1u == gridDim.x * gridDim.y) {
            __hip_atomic_store(F.blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int k = 0; k < F.n; ++k) st_sys_u32(F.f[k], F.seq);
        }
    }
}

// ---- checkpoints of the restartable state (round 5: in-place recovery of a group after a timed-out wait of the direct transport) --------------------
// What survives a step: the particles (position + three APIC rows), the two pressure volumes (warm starts), the counts.  A generation
// is only ever written while the slab's time-out mark is clear, so every generation a slab holds was taken before its first failed wait.
struct SlabCheckpoint { float4* part[4]; float* pressure[2]; uint32_t* n; uint32_t* step; };      // device buffers of ONE generation
__global__ __launch_bounds__(256) void k_slab_checkpoint(SlabCheckpoint ck, const float4* __restrict__ pos, const float4* __restrict__ vx, const float4* __restrict__ vy,
                                                         const float4* __restrict__ vz, const float* __restrict__ p0, const float* __restrict__ p1, uint32_t vol_quads,
                                                         const uint32_t* __restrict__ n_dev, uint32_t n_host, const uint32_t* __restrict__ dir_error, uint32_t step) {
    if (dir_error && __hip_atomic_load(dir_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // (uniform) never overwrite a good generation with a doubtful state
    const uint32_t n = n_dev ? n_dev[0] : n_host;
    const float4* src[4] = {pos, vx, vy, vz};
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ck.part[k][i] = src[k][i];
    }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < vol_quads; i += gridDim.x * 256) {
        reinterpret_cast<float4*>(ck.pressure[0])[i] = reinterpret_cast<const float4*>(p0)[i];
        reinterpret_cast<float4*>(ck.pressure[1])[i] = reinterpret_cast<const float4*>(p1)[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { ck.n[0] = n; *ck.step = step; }
}
__global__ __launch_bounds__(256) void k_slab_restore(SlabCheckpoint ck, float4* __restrict__ pos, float4* __restrict__ vx, float4* __restrict__ vy, float4* __restrict__ vz,
                                                      float* __restrict__ p0, float* __restrict__ p1, uint32_t vol_quads, uint32_t* __restrict__ n_dev) {
    const uint32_t n = ck.n[0];
    float4* dst[4] = {pos, vx, vy, vz};
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k][i] = ck.part[k][i];
    }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < vol_quads; i += gridDim.x * 256) {
        reinterpret_cast<float4*>(p0)[i] = reinterpret_cast<const float4*>(ck.pressure[0])[i];
        reinterpret_cast<float4*>(p1)[i] = reinterpret_cast<const float4*>(ck.pressure[1])[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_dev) { n_dev[0] = n; n_dev[1] = 0u; n_dev[2] = 0u; }
}

// FLUID bricks of the own range per brick layer (blub_slab_group_rebalance): hist[layer] += 1 for every entry of the fluid list (hist zeroed by the caller)
__global__ __launch_bounds__(256) void k_slab_layer_histogram(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, float* __restrict__ hist) {
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int bx, by, bz; brick_coords(bg, list[i], bx, by, bz);
        atomicAdd(hist + bz, 1.0f);      // (integers below 2^24: exact, order-independent)
    }
}

// Dot products across slabs: every slab's PCG kernels write their per-block partials into segment `rank` of a gather array
// of nranks x SLAB_NP entries; after the segments have been exchanged (p2p, see slab_gather) the unchanged consumer kernels
// re-reduce all nranks x SLAB_NP partials in the same fixed order on every slab => identical scalars and identical
// convergence decisions everywhere, no all-reduce, no extra reduction kernels.
// PCG grid (= partials per slab) of a slab solve: the same on every slab (the gathered segments have one size), chosen per step from the
// largest fluid-brick count of any slab (gathered at the start of the step, blub_slab.inc.hip: slab_step)
constexpr int SLAB_NP_MAX = 1024, SLAB_NP_DEFAULT = 512;

}  // namespace blubk
