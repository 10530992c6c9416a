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

struct SlabCounts { uint32_t n_stay, n_up, n_down, n_leave, n_holes, n_fill, pad0, pad1; };   // zeroed before every exchange

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
                                                     float4* __restrict__ out_pos, float4* __restrict__ out_vx, float4* __restrict__ out_vy, float4* __restrict__ out_vz) {
    __shared__ uint32_t wcount[4], wbase;
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
                                                           uint32_t* __restrict__ leave_idx) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) p = pos[i];
    const bool up = live && p.z >= z1, down = live && p.z < z0;
    if (__ballot(up || down) == 0ull) return;          // the usual case: nobody in this wave leaves
    const uint32_t ku = wave_alloc(&counts->n_up, up), kd = wave_alloc(&counts->n_down, down), kl = wave_alloc(&counts->n_leave, up || down);
    if (up) { if (ku < capacity) { up_pos[ku] = p; up_vx[ku] = vx[i]; up_vy[ku] = vy[i]; up_vz[ku] = vz[i]; } }
    else if (down) { if (kd < capacity) { dn_pos[kd] = p; dn_vx[kd] = vx[i]; dn_vy[kd] = vy[i]; dn_vz[kd] = vz[i]; } }
    if (up || down) leave_idx[kl] = i;
}
__global__ __launch_bounds__(256) void k_slab_migrate_match(uint32_t n, const float4* __restrict__ pos, float z0, float z1, SlabCounts* __restrict__ counts,
                                                            const uint32_t* __restrict__ leave_idx, uint32_t* __restrict__ hole_idx, uint32_t* __restrict__ fill_idx) {
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
                                                                    uint32_t* __restrict__ heads) {
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

// loopback transport: all plane copies of one halo exchange in ONE launch (a hipMemcpyAsync per plane costs ~4 us of queue time each,
// ~100 of them per step)
struct SlabCopy { const void* src; void* dst; uint32_t bytes; uint32_t pad; };   // src, dst 16-byte aligned, bytes % 16 == 0
constexpr int SLAB_COPY_MAX = 40;
struct SlabCopyList { SlabCopy c[SLAB_COPY_MAX]; int n; };
__global__ __launch_bounds__(256) void k_slab_copy_planes(SlabCopyList L) {
    const SlabCopy c = L.c[blockIdx.y];
    const uint32_t n16 = c.bytes >> 4;
    const uint4* __restrict__ s = reinterpret_cast<const uint4*>(c.src);
    uint4* __restrict__ d = reinterpret_cast<uint4*>(c.dst);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) d[i] = s[i];
}

// Dot products across slabs: every slab's PCG kernels write their per-block partials into segment `rank` of a gather array
// of nranks x SLAB_NP entries; after the segments have been exchanged (p2p, see slab_gather) the unchanged consumer kernels
// re-reduce all nranks x SLAB_NP partials in the same fixed order on every slab => identical scalars and identical
// convergence decisions everywhere, no all-reduce, no extra reduction kernels.
// PCG grid (= partials per slab) of a slab solve: the same on every slab (the gathered segments have one size), chosen per step from the
// largest fluid-brick count of any slab (gathered at the start of the step, blub_slab.inc.hip: slab_step)
constexpr int SLAB_NP_MAX = 1024, SLAB_NP_DEFAULT = 512;

}  // namespace blubk
