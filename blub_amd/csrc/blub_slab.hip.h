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

struct SlabCounts { uint32_t n_stay, n_up, n_down, pad; };

// Copies (does not remove) the own particles within [zlo, zhi) into a send buffer; rows == nullptr => positions only.
__global__ __launch_bounds__(256) void k_slab_select(uint32_t n, const float4* __restrict__ pos, const float4* __restrict__ vx, const float4* __restrict__ vy,
                                                     const float4* __restrict__ vz, float zlo, float zhi, uint32_t capacity, uint32_t* __restrict__ counter,
                                                     float4* __restrict__ out_pos, float4* __restrict__ out_vx, float4* __restrict__ out_vy, float4* __restrict__ out_vz) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos[i];
    if (!(p.z >= zlo && p.z < zhi)) return;
    const uint32_t k = atomicAdd(counter, 1u);
    if (k >= capacity) return;
    out_pos[k] = p;
    if (out_vx) { out_vx[k] = vx[i]; out_vy[k] = vy[i]; out_vz[k] = vz[i]; }
}

// Migration: particles that left [z0, z1) go to the up / down send buffers, the rest is compacted into the *_new arrays.
__global__ __launch_bounds__(256) void k_slab_partition(uint32_t n, const float4* __restrict__ pos, const float4* __restrict__ vx, const float4* __restrict__ vy,
                                                        const float4* __restrict__ vz, float z0, float z1, uint32_t capacity, SlabCounts* __restrict__ counts,
                                                        float4* __restrict__ pos_new, float4* __restrict__ vx_new, float4* __restrict__ vy_new, float4* __restrict__ vz_new,
                                                        float4* __restrict__ up_pos, float4* __restrict__ up_vx, float4* __restrict__ up_vy, float4* __restrict__ up_vz,
                                                        float4* __restrict__ dn_pos, float4* __restrict__ dn_vx, float4* __restrict__ dn_vy, float4* __restrict__ dn_vz) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos[i];
    if (p.z >= z1) {
        const uint32_t k = atomicAdd(&counts->n_up, 1u);
        if (k < capacity) { up_pos[k] = p; up_vx[k] = vx[i]; up_vy[k] = vy[i]; up_vz[k] = vz[i]; }
    } else if (p.z < z0) {
        const uint32_t k = atomicAdd(&counts->n_down, 1u);
        if (k < capacity) { dn_pos[k] = p; dn_vx[k] = vx[i]; dn_vy[k] = vy[i]; dn_vz[k] = vz[i]; }
    } else {
        const uint32_t k = atomicAdd(&counts->n_stay, 1u);
        pos_new[k] = p; vx_new[k] = vx[i]; vy_new[k] = vy[i]; vz_new[k] = vz[i];
    }
}

// Ghost particles for the density projection: mark their cells FLUID and hang them into the density linked list
// (what advect_particles.comp:176-181 does for the own particles).
__global__ __launch_bounds__(256) void k_slab_insert_density_ghosts(Grid g, uint32_t first, uint32_t count, float4* __restrict__ pos, int8_t* __restrict__ marker,
                                                                    uint32_t* __restrict__ heads) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const uint32_t i = first + k;
    const float4 p = pos[i];
    {
        const int x = (int)p.x, y = (int)p.y, z = (int)p.z;
        if (inb(g, x, y, z)) { const int c = cidx(g, x, y, z); if (marker[c] != CELL_SOLID) marker[c] = CELL_FLUID; }
    }
    uint32_t old = 0;
    const int dx = (int)(p.x - 0.5f), dy = (int)(p.y - 0.5f), dz = (int)(p.z - 0.5f);
    if (inb(g, dx, dy, dz)) old = atomicExch(heads + cidx(g, dx, dy, dz), i + 1);
    reinterpret_cast<uint32_t*>(pos)[4 * (size_t)i + 3] = old - 1u;
}

// Dot products across slabs: every slab's PCG kernels write their per-block partials into segment `rank` of a gather array
// of nranks x SLAB_NP entries; after the segments have been exchanged (p2p, see slab_gather) the unchanged consumer kernels
// re-reduce all nranks x SLAB_NP partials in the same fixed order on every slab => identical scalars and identical
// convergence decisions everywhere, no all-reduce, no extra reduction kernels.
constexpr int SLAB_NP = 256;   // PCG grid (= partials per slab) of a slab solve
struct SlabPtrs { float* p[8]; };
// loopback transport: copy segment s of slab s's array into every other local slab's array
__global__ __launch_bounds__(256) void k_slab_gather_local(SlabPtrs ptrs, int nslabs, int seg_floats) {
    const int s = blockIdx.x;
    for (int j = threadIdx.x; j < seg_floats; j += 256) {
        const float v = ptrs.p[s][(size_t)s * seg_floats + j];
        for (int d = 0; d < nslabs; ++d) if (d != s) ptrs.p[d][(size_t)s * seg_floats + j] = v;
    }
}

}  // namespace blubk
