// Brick-sparse work decomposition for the grid kernels (MI355X-first redesign of the reference's dense dispatches).
//
// The reference dispatches every grid shader over all N cells (hybrid_fluid.rs:786).  At the headline operating point
// (1 M particles in 256^3) < 1 % of the cells are FLUID, so here the volumes stay dense in HBM (the renderer contract,
// hybrid_fluid.rs:700-721, wants plain volumes) but the WORK is driven by compact lists of 16x8x4-cell bricks:
//   fluid list  : bricks that contain a particle (= contain a FLUID cell)
//   active list : fluid bricks dilated by one brick (every cell within >= 4 cells of the fluid; all stencils of the
//                 step reach at most 2 cells beyond a FLUID cell)
//   reset list  : active bricks + "stale" bricks (touched by an earlier step, not active any more) whose velocity /
//                 pressure / marker contents are put back to the state the reference's dense passes would leave there
//                 (velocity 0: divergence_remove.comp:36-38, pressure 0: pressure_init.comp:45-48, marker AIR/SOLID).
// Invariant: outside the bricks ever touched, every volume holds exactly what the dense reference would hold.
// Lists are rebuilt on the device twice per step (before P2G, after advection) from the particle positions with a
// deterministic scan, so kernels are launched with a fixed grid and loop `for (i = blockIdx.x; i < *count; ...)`.
#pragma once
#include "blub_kernels.hip.h"

namespace blubk {

constexpr int BX = 16, BY = 8, BZ = 4;        // brick extent in cells (x fastest): 512 cells = 128 quads
static_assert(BX == 16 && BY == 8 && BZ == 4, "k_advect (blub_kernels.hip.h) marks FLUID bricks with these extents as shifts");
constexpr int BRICK_THREADS = 128;            // one thread per quad (4 x-consecutive cells)
constexpr uint32_t STALE_BIT = 0x80000000u;

struct BrickGeom {
    Grid g;
    int nbx, nby, nbz, nb;
    uint32_t mx, my;      // ceil(2^32 / nbx), ceil(2^32 / nby): brick index -> coordinates without integer division (brick_coords)
};
struct BrickCounts {        // device resident; `seq` tags the asynchronous host read-back of this list build
    uint32_t n_fluid, n_active, n_reset, n_stale;
    uint32_t seq, pad0, pad1, seq_check;
};

__device__ __forceinline__ uint32_t brick_of_cell(const BrickGeom& bg, int x, int y, int z) {
    return (uint32_t)(((z / BZ) * bg.nby + (y / BY)) * bg.nbx + (x / BX));
}
// b = (bz * nby + by) * nbx + bx  ->  (bx, by, bz).  A 32-bit division by a run-time value costs ~20 VALU instructions and the
// latency-bound brick kernels do three per brick; with m = ceil(2^32 / d), umulhi(n, m) == n / d whenever n * d < 2^32
// (n < nb: checked when the handle is created), d == 1 excepted (its multiplier does not fit).
__device__ __forceinline__ void brick_coords(const BrickGeom& bg, uint32_t b, int& bx, int& by, int& bz) {
    const uint32_t q1 = bg.nbx == 1 ? b : __umulhi(b, bg.mx);
    const uint32_t q2 = bg.nby == 1 ? q1 : __umulhi(q1, bg.my);
    bx = (int)(b - q1 * (uint32_t)bg.nbx); by = (int)(q1 - q2 * (uint32_t)bg.nby); bz = (int)q2;
}
inline void brick_geom_set_magic(BrickGeom& bg) {    // host side
    bg.mx = bg.nbx > 1 ? (uint32_t)((0x100000000ull + (uint32_t)bg.nbx - 1) / (uint32_t)bg.nbx) : 0u;
    bg.my = bg.nby > 1 ? (uint32_t)((0x100000000ull + (uint32_t)bg.nby - 1) / (uint32_t)bg.nby) : 0u;
}
// thread -> quad of brick b; returns false if the quad lies outside the grid
__device__ __forceinline__ bool brick_quad(const BrickGeom& bg, uint32_t b, int t, int& x0, int& y, int& z) {
    int bx, by, bz; brick_coords(bg, b, bx, by, bz);
    x0 = bx * BX + ((t & 3) << 2);
    y = by * BY + ((t >> 2) & 7);
    z = bz * BZ + (t >> 5);
    return x0 < bg.g.nx && y < bg.g.ny && z < bg.g.nz;
}

// XCD-contiguous order for kernels that loop over a brick list: workgroup b runs on XCD b % 8 (observed dispatch order, a speed hint only -- any
// placement is correct), and every kernel starts with cold L2s, so list position j -> (j % 8) * ceil(n / 8) + j / 8 hands XCD k the contiguous
// range [k n / 8, (k + 1) n / 8) of the list: x-neighbouring bricks (consecutive in the list, sharing 64-byte sectors of every row they touch and
// each other's halo cells) meet in ONE L2 instead of being fetched by up to four.  Loop j over [0, list_slots(n)) with a stride that is a multiple of 8.
__device__ __forceinline__ uint32_t list_slots(uint32_t n) { return (n + 7u) & ~7u; }
__device__ __forceinline__ bool list_slot(uint32_t j, uint32_t n, uint32_t& i) { i = (j & 7u) * ((n + 7u) >> 3) + (j >> 3); return i < n; }

// ---- list construction ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bricks_mark_particles(BrickGeom bg, uint32_t num_particles, const float4* __restrict__ pos, uint8_t* __restrict__ brick_fluid,
                                                               const uint32_t* __restrict__ n_dev, uint32_t n_sel) {
    num_particles = particle_count(num_particles, n_dev, n_sel);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_particles) return;
    const float4 p = pos[i];
    const int x = (int)p.x, y = (int)p.y, z = (int)p.z;
    if (inb(bg.g, x, y, z)) brick_fluid[brick_of_cell(bg, x, y, z)] = 1;
}
// stand-alone stage calls (test hook): derive the fluid bricks from the marker volume instead of the particles
__global__ __launch_bounds__(BRICK_THREADS) void k_bricks_mark_from_marker(BrickGeom bg, const int8_t* __restrict__ marker, uint8_t* __restrict__ brick_fluid) {
    const uint32_t b = blockIdx.x;
    int x0, y, z;
    bool any = false;
    if (brick_quad(bg, b, threadIdx.x, x0, y, z)) any = any_fluid4(*reinterpret_cast<const uint32_t*>(marker + cidx(bg.g, x0, y, z)));
    if (__syncthreads_or(any) && threadIdx.x == 0) brick_fluid[b] = 1;
}

__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* sm /*17*/, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint32_t s = sm[w]; if (w < wave) woff += s; tot += s; }
    total = tot;
    return woff + inc - v;
}

enum { COMPACT_STEP_A = 0, COMPACT_STEP_B = 1, COMPACT_ALL_ACTIVE = 2 };
enum { BF_FLUID = 1, BF_ACTIVE = 2, BF_STALE = 4, BF_RESET = 8 };   // FLUID / ACTIVE only for bricks of the own z-slab
// Pass 1 (one thread per brick, 1024 bricks per block): dilate the fluid flags, classify each brick, count per block.
__global__ __launch_bounds__(1024) void k_bricks_classify(BrickGeom bg, int phase, int all_touched, int own_bz_lo, int own_bz_hi, const uint8_t* __restrict__ brick_fluid,
                                                          uint8_t* __restrict__ brick_active, uint8_t* __restrict__ brick_touched,
                                                          uint8_t* __restrict__ brick_flags, uint4* __restrict__ block_counts) {
    __shared__ uint32_t sm[17];
    const int b = blockIdx.x * 1024 + threadIdx.x;
    uint32_t fl = 0;
    if (b < bg.nb) {
        const bool f = brick_fluid[b] != 0;
        bool act;
        if (phase == COMPACT_ALL_ACTIVE) act = true;
        else {
            int bx, by, bz; brick_coords(bg, b, bx, by, bz);
            uint32_t any = 0;   // all 27 flags are loaded unconditionally (one batch of independent loads, no short-circuit chain)
#pragma unroll
            for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int qx = bx + dx, qy = by + dy, qz = bz + dz;
                        const bool inside = (unsigned)qx < (unsigned)bg.nbx && (unsigned)qy < (unsigned)bg.nby && (unsigned)qz < (unsigned)bg.nbz;
                        any |= (uint32_t)brick_fluid[inside ? (qz * bg.nby + qy) * bg.nbx + qx : b] & (inside ? 0xFFu : 0u);
                    }
            act = any != 0;
            if (phase == COMPACT_STEP_B) act = act || brick_active[b] != 0;
        }
        const bool touched = all_touched || brick_touched[b] != 0;
        const bool stale = (phase == COMPACT_STEP_A) && touched && !act;
        int bx_own, by_own, bz_own; brick_coords(bg, b, bx_own, by_own, bz_own); (void)bx_own; (void)by_own;
        const bool own = bz_own >= own_bz_lo && bz_own < own_bz_hi;   // z-slab decomposition: lists of work hold own bricks only
        fl = ((f && own) ? BF_FLUID : 0) | ((act && own) ? BF_ACTIVE : 0) | (stale ? BF_STALE : 0) | ((act || stale) ? BF_RESET : 0);
        brick_flags[b] = (uint8_t)fl;
        brick_active[b] = act;
        if (phase == COMPACT_STEP_A) brick_touched[b] = act;
        else if (act) brick_touched[b] = 1;
    }
    // per-block totals of the four flags: wave ballots + one LDS exchange (no scans needed here)
    const uint32_t packed = (uint32_t)__popcll(__ballot((fl & BF_FLUID) != 0)) | ((uint32_t)__popcll(__ballot((fl & BF_ACTIVE) != 0)) << 8) |
                            ((uint32_t)__popcll(__ballot((fl & BF_RESET) != 0)) << 16) | ((uint32_t)__popcll(__ballot((fl & BF_STALE) != 0)) << 24);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = packed;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tf = 0, ta = 0, tr = 0, ts = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t q = sm[w]; tf += q & 0xFFu; ta += (q >> 8) & 0xFFu; tr += (q >> 16) & 0xFFu; ts += q >> 24; }
        block_counts[blockIdx.x] = make_uint4(tf, ta, tr, ts);
    }
}
// Pass 2: every block adds up the counts of the blocks before it (<= 256 of them) and scatters its bricks: the lists
// come out in brick (= memory) order, deterministically.
__global__ __launch_bounds__(1024) void k_bricks_scatter(BrickGeom bg, const uint8_t* __restrict__ brick_flags, const uint4* __restrict__ block_counts, int nblocks,
                                                         uint32_t* __restrict__ list_fluid, uint32_t* __restrict__ list_active,
                                                         uint32_t* __restrict__ list_reset, BrickCounts* __restrict__ counts, uint32_t seq,
                                                         uint8_t* __restrict__ brick_fluid_to_clear, BrickCounts* __restrict__ host_snapshot, uint32_t* __restrict__ resort_cursor) {
    __shared__ uint32_t sm[17];
    __shared__ uint32_t base[4], total[4];
    if (resort_cursor && blockIdx.x == 0 && threadIdx.x == 0) *resort_cursor = 0;      // bump allocator of the particle re-sort (k_resort_scan): empty after every list build
    if (threadIdx.x < 64) {   // one wave sums the block counts: lanes stride over the blocks in order
        uint32_t before[4] = {0, 0, 0, 0}, all[4] = {0, 0, 0, 0};
        for (int k = threadIdx.x; k < nblocks; k += 64) {
            const uint4 c = block_counts[k];
            all[0] += c.x; all[1] += c.y; all[2] += c.z; all[3] += c.w;
            if (k < (int)blockIdx.x) { before[0] += c.x; before[1] += c.y; before[2] += c.z; before[3] += c.w; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { before[q] += __shfl_down(before[q], off, 64); all[q] += __shfl_down(all[q], off, 64); }
        if (threadIdx.x == 0) for (int q = 0; q < 4; ++q) { base[q] = before[q]; total[q] = all[q]; }
    }
    __syncthreads();
    const int b = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t fl = b < bg.nb ? brick_flags[b] : 0u;
    if (b < bg.nb && brick_fluid_to_clear) brick_fluid_to_clear[b] = 0;   // consumed by k_bricks_classify: ready for the next build
    // exclusive prefix of the three flags: inside a wave = popcount of the ballot below the lane, across waves = one LDS
    // exchange of the (packed) wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long bf = __ballot((fl & BF_FLUID) != 0), ba = __ballot((fl & BF_ACTIVE) != 0), br = __ballot((fl & BF_RESET) != 0);
    __shared__ uint32_t wtot[16];
    if (lane == 0) wtot[wave] = (uint32_t)__popcll(bf) | ((uint32_t)__popcll(ba) << 8) | ((uint32_t)__popcll(br) << 16);
    __syncthreads();
    uint32_t of = (uint32_t)__popcll(bf & below), oa = (uint32_t)__popcll(ba & below), orr = (uint32_t)__popcll(br & below);
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint32_t q = w < wave ? wtot[w] : 0u; of += q & 0xFFu; oa += (q >> 8) & 0xFFu; orr += (q >> 16) & 0xFFu; }
    (void)sm;
    if (fl & BF_FLUID) list_fluid[base[0] + of] = (uint32_t)b;
    if (fl & BF_ACTIVE) list_active[base[1] + oa] = (uint32_t)b;
    if (fl & BF_RESET) list_reset[base[2] + orr] = (uint32_t)b | ((fl & BF_STALE) ? STALE_BIT : 0u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        BrickCounts c; c.n_fluid = total[0]; c.n_active = total[1]; c.n_reset = total[2]; c.n_stale = total[3]; c.seq = seq; c.pad0 = 0; c.pad1 = 0; c.seq_check = seq;
        *counts = c;
        if (host_snapshot) {   // pinned host ring slot (path selection only): payload first, tags last
            host_snapshot->n_fluid = c.n_fluid; host_snapshot->n_active = c.n_active; host_snapshot->n_reset = c.n_reset; host_snapshot->n_stale = c.n_stale;
            __threadfence_system();
            host_snapshot->seq = seq; host_snapshot->seq_check = seq;
        }
    }
}

// Passes 1 + 2 in ONE launch (the headline scene is bound by its number of dependent launches: two list builds per step).  Every block
// classifies its 1024 bricks, publishes its four counts followed by a sequence-tagged ready flag (release), waits until the flags of ALL
// blocks carry this build's sequence number (every block of the grid is co-resident: the host only uses this kernel while the grid has at
// most one block per CU) and scatters with the same deterministic offsets as k_bricks_scatter.  Lists, counts and snapshot are identical
// to the two-kernel build.
__global__ __launch_bounds__(1024) void k_bricks_build(BrickGeom bg, int phase, int all_touched, int own_bz_lo, int own_bz_hi, uint8_t* __restrict__ brick_fluid,
                                                       uint8_t* __restrict__ brick_active, uint8_t* __restrict__ brick_touched, uint32_t* block_counts4 /* 4 per block */,
                                                       uint32_t* block_ready, uint32_t* __restrict__ list_fluid, uint32_t* __restrict__ list_active,
                                                       uint32_t* __restrict__ list_reset, BrickCounts* __restrict__ counts, uint32_t seq, BrickCounts* __restrict__ host_snapshot,
                                                       uint32_t* __restrict__ sticky_timeout, uint32_t* __restrict__ resort_cursor) {
    __shared__ uint32_t sm[17];
    __shared__ uint32_t base[4], total[4];
    __shared__ uint32_t wtot[16];
    __shared__ int timed_out;
    if (resort_cursor && blockIdx.x == 0 && threadIdx.x == 0) *resort_cursor = 0;      // bump allocator of the particle re-sort (k_resort_scan): empty after every list build
    const int b = blockIdx.x * 1024 + threadIdx.x;
    const int nblocks = gridDim.x;
    uint32_t fl = 0;
    if (b < bg.nb) {
        const bool f = brick_fluid[b] != 0;
        bool act;
        if (phase == COMPACT_ALL_ACTIVE) act = true;
        else {
            int bx, by, bz; brick_coords(bg, b, bx, by, bz);
            uint32_t any = 0;   // all 27 flags are loaded unconditionally (one batch of independent loads, no short-circuit chain)
#pragma unroll
            for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int qx = bx + dx, qy = by + dy, qz = bz + dz;
                        const bool inside = (unsigned)qx < (unsigned)bg.nbx && (unsigned)qy < (unsigned)bg.nby && (unsigned)qz < (unsigned)bg.nbz;
                        any |= (uint32_t)brick_fluid[inside ? (qz * bg.nby + qy) * bg.nbx + qx : b] & (inside ? 0xFFu : 0u);
                    }
            act = any != 0;
            if (phase == COMPACT_STEP_B) act = act || brick_active[b] != 0;
        }
        const bool touched = all_touched || brick_touched[b] != 0;
        const bool stale = (phase == COMPACT_STEP_A) && touched && !act;
        int bx_own, by_own, bz_own; brick_coords(bg, b, bx_own, by_own, bz_own); (void)bx_own; (void)by_own;
        const bool own = bz_own >= own_bz_lo && bz_own < own_bz_hi;   // z-slab decomposition: lists of work hold own bricks only
        fl = ((f && own) ? BF_FLUID : 0) | ((act && own) ? BF_ACTIVE : 0) | (stale ? BF_STALE : 0) | ((act || stale) ? BF_RESET : 0);
        brick_active[b] = act;
        if (phase == COMPACT_STEP_A) brick_touched[b] = act;
        else if (act) brick_touched[b] = 1;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bf = __ballot((fl & BF_FLUID) != 0), ba = __ballot((fl & BF_ACTIVE) != 0), br = __ballot((fl & BF_RESET) != 0), bs = __ballot((fl & BF_STALE) != 0);
    if (lane == 0) { wtot[wave] = (uint32_t)__popcll(bf) | ((uint32_t)__popcll(ba) << 8) | ((uint32_t)__popcll(br) << 16); sm[wave] = (uint32_t)__popcll(bs); }
    __syncthreads();      // (also: every thread of the block has read its 27 neighbour flags before any of them is cleared below -- for THIS block's bricks;
                          //  other blocks' flags are cleared by their owners only after the grid-wide wait that follows)
    // A block's four counts (<= 1024 each: 11 bits) and the build's tag travel in ONE 64-bit word: the store publishes them atomically, so neither
    // a release fence (an L2 write-back) on this side nor an acquire fence and a second load on the waiting side are needed.
    unsigned long long* const slots = reinterpret_cast<unsigned long long*>(block_counts4);      // 16 bytes per block, the first 8 used here
    const unsigned long long tag = (unsigned long long)(seq % 0xFFFFFu) + 1ull;                    // 1 .. 2^20 - 1: never the 0 of fresh memory
    if (threadIdx.x == 0) {
        uint32_t tf = 0, ta = 0, tr = 0, ts = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t q = wtot[w]; tf += q & 0xFFu; ta += (q >> 8) & 0xFFu; tr += (q >> 16) & 0xFFu; ts += sm[w]; }
        const unsigned long long word = (tag << 44) | (unsigned long long)tf | ((unsigned long long)ta << 11) | ((unsigned long long)tr << 22) | ((unsigned long long)ts << 33);
        __hip_atomic_store(slots + 2 * blockIdx.x, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) timed_out = 0;
    if (threadIdx.x < 64) {   // one wave waits for all blocks and sums their counts: lanes stride over the blocks in order
        uint32_t before[4] = {0, 0, 0, 0}, all[4] = {0, 0, 0, 0};
        for (int k = threadIdx.x; k < nblocks; k += 64) {
            // (bounded: if the blocks were NOT co-resident after all -- a masked or partitioned device -- the build ends with a wrong list and the
            //  counts carry an error mark instead of hanging the GPU; the host then reports BLUB_ERR_DEVICE and falls back to the two-kernel build)
            unsigned spins = 0;
            unsigned long long word;
            while (((word = __hip_atomic_load(slots + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 44) != tag) { __builtin_amdgcn_s_sleep(2); if (++spins > (1u << 24)) { timed_out = 1; word = 0; if (sticky_timeout) *sticky_timeout = 1u; break; } }      // (sticky: a pinned word no later build overwrites)
            const uint32_t c[4] = {(uint32_t)(word & 0x7FFu), (uint32_t)((word >> 11) & 0x7FFu), (uint32_t)((word >> 22) & 0x7FFu), (uint32_t)((word >> 33) & 0x7FFu)};
#pragma unroll
            for (int q = 0; q < 4; ++q) { all[q] += c[q]; if (k < (int)blockIdx.x) before[q] += c[q]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { before[q] += __shfl_down(before[q], off, 64); all[q] += __shfl_down(all[q], off, 64); }
        if (threadIdx.x == 0) for (int q = 0; q < 4; ++q) { base[q] = before[q]; total[q] = all[q]; }
    }
    __syncthreads();
    // every block has classified by now (its flag is set after its classification): the marks are consumed, clear them for the next build
    if (b < bg.nb) brick_fluid[b] = 0;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t of = (uint32_t)__popcll(bf & below), oa = (uint32_t)__popcll(ba & below), orr = (uint32_t)__popcll(br & below);
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint32_t q = w < wave ? wtot[w] : 0u; of += q & 0xFFu; oa += (q >> 8) & 0xFFu; orr += (q >> 16) & 0xFFu; }
    if (fl & BF_FLUID) list_fluid[base[0] + of] = (uint32_t)b;
    if (fl & BF_ACTIVE) list_active[base[1] + oa] = (uint32_t)b;
    if (fl & BF_RESET) list_reset[base[2] + orr] = (uint32_t)b | ((fl & BF_STALE) ? STALE_BIT : 0u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        BrickCounts c; c.n_fluid = total[0]; c.n_active = total[1]; c.n_reset = total[2]; c.n_stale = total[3]; c.seq = seq; c.pad0 = (uint32_t)timed_out; c.pad1 = 0; c.seq_check = seq;
        *counts = c;
        if (host_snapshot) {   // pinned host ring slot (path selection only): payload first, tags last
            host_snapshot->n_fluid = c.n_fluid; host_snapshot->n_active = c.n_active; host_snapshot->n_reset = c.n_reset; host_snapshot->n_stale = c.n_stale;
            host_snapshot->pad0 = c.pad0;
            __threadfence_system();
            host_snapshot->seq = seq; host_snapshot->seq_check = seq;
        }
    }
}

// ---- internal re-sort of the particles (round 6) ---------------------------------------------------------------------------------------------
// Every particle kernel is bound by 64-byte sector requests, and how many a wave makes depends on the ORDER of the particles in memory: 60 steps
// after a rebinning the list build / P2G walk / advection / density walk / correction cost 108 / 90 / 51 / 38 / 34 us against 31 / 55 / 31 / 14 / 16 us
// right after it (profiles/r05_kernel_stats_v2_sparse_bench.csv, min .. max).  The reference rebins every 60 steps because its binning is three dense
// passes; here a brick-sparse counting sort at the SAME point of the step (between the velocity solve and the projection, where the velocity rows
// are dead and only positions move: hybrid_fluid.rs:857-893) costs three small launches, so the engine re-sorts every few steps on its own:
//   count : key = (brick, cell in brick); rank by wave-aggregated atomic adds on a brick-major counter table
//   scan  : one workgroup per FLUID brick turns its 512 counters into offsets behind a base from a bump allocator and zeroes them
//   move  : position and particle id to offset + rank
// The order the CALLER sees (blub_fluid_get_particles, the reference's rebinning cadence) is kept by the particle ids: internal slot s holds
// the particle with canonical index pid[s]; every entry point that exposes an order restores it first (restore_canonical_order, blub_fluid.hip).
__device__ __forceinline__ int resort_key(const BrickGeom& bg, float px, float py, float pz) {
    const int x = (int)px, y = (int)py, z = (int)pz;
    if (!inb(bg.g, x, y, z)) return bg.nb * (BX * BY * BZ);                               // one bucket for whatever lies outside the grid
    return (int)brick_of_cell(bg, x, y, z) * (BX * BY * BZ) + (((z % BZ) * BY + (y % BY)) * BX + (x % BX));
}
__global__ __launch_bounds__(256) void k_resort_count(BrickGeom bg, uint32_t num_particles, const float4* __restrict__ pos, uint32_t* __restrict__ counters, uint32_t* __restrict__ ranks) {
    if (blockIdx.x * 256u >= num_particles) return;      // (uniform)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < num_particles;                 // no early return: the wave-level insertion needs every lane
    const float4 p = live ? pos[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int key = live ? resort_key(bg, p.x, p.y, p.z) : -1;
    const unsigned long long mine = wave_group(key);
    const uint32_t r = wave_group_rank(wave_group_add(counters, key, mine), mine);
    if (live) ranks[i] = r;
}
constexpr int RESORT_GROUP = 4;      // FLUID bricks per workgroup and allocation: the allocator is ONE address (~11 ns per atomic, MI355X_MICROARCH.md "dequeue")
__global__ __launch_bounds__(BRICK_THREADS) void k_resort_scan(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                               uint32_t* __restrict__ counters, uint32_t* __restrict__ starts, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t wave_tot[RESORT_GROUP][2], base_of[RESORT_GROUP];
    const uint32_t n = *count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i0 = blockIdx.x * RESORT_GROUP; i0 <= n; i0 += gridDim.x * RESORT_GROUP) {
        uint4 q[RESORT_GROUP]; uint32_t s[RESORT_GROUP], inc[RESORT_GROUP]; size_t at[RESORT_GROUP];
#pragma unroll
        for (int k = 0; k < RESORT_GROUP; ++k) {
            const bool valid = i0 + k < n;
            at[k] = valid ? (size_t)list[i0 + k] * (BX * BY * BZ) + 4 * threadIdx.x : 0;
            q[k] = valid ? *reinterpret_cast<const uint4*>(counters + at[k]) : make_uint4(0, 0, 0, 0);
            s[k] = q[k].x + q[k].y + q[k].z + q[k].w;
            inc[k] = wave_inclusive_scan(s[k]);
            if (lane == 63) wave_tot[k][wave] = inc[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot[RESORT_GROUP], sum = 0;
#pragma unroll
            for (int k = 0; k < RESORT_GROUP; ++k) { tot[k] = wave_tot[k][0] + wave_tot[k][1]; sum += tot[k]; }
            const bool outside_here = n - i0 < (uint32_t)RESORT_GROUP;       // this group holds list position n: it also places the bucket outside the grid
            const uint32_t ko = (uint32_t)bg.nb * (BX * BY * BZ), co = outside_here ? counters[ko] : 0u;
            uint32_t b0 = (sum + co) ? atomicAdd(cursor, sum + co) : 0u;
#pragma unroll
            for (int k = 0; k < RESORT_GROUP; ++k) { base_of[k] = b0; b0 += tot[k]; }
            if (outside_here) { starts[ko] = b0; counters[ko] = 0; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RESORT_GROUP; ++k) {
            if (i0 + k >= n) continue;
            const uint32_t f0 = base_of[k] + (wave ? wave_tot[k][0] : 0u) + inc[k] - s[k];
            *reinterpret_cast<uint4*>(starts + at[k]) = make_uint4(f0, f0 + q[k].x, f0 + q[k].x + q[k].y, f0 + q[k].x + q[k].y + q[k].z);
            if (s[k]) *reinterpret_cast<uint4*>(counters + at[k]) = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();      // (the next group of this workgroup reuses wave_tot / base_of)
    }
}
__global__ __launch_bounds__(256) void k_resort_move(BrickGeom bg, uint32_t num_particles, const float4* __restrict__ pos, const uint32_t* __restrict__ pid,
                                                     const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ starts, float4* __restrict__ pos_out, uint32_t* __restrict__ pid_out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_particles) return;
    const float4 p = pos[i];
    const uint32_t dst = starts[resort_key(bg, p.x, p.y, p.z)] + ranks[i];
    if (dst < num_particles) { pos_out[dst] = p; pid_out[dst] = pid[i]; }      // (always: the counts are those of these positions)
}
// canonical order back: record pid[s] <- internal slot s; links (particles_position_ll.w, list heads) are particle indices and follow
__global__ __launch_bounds__(256) void k_unpermute(uint32_t n, const uint32_t* __restrict__ pid, const float4* __restrict__ in, float4* __restrict__ out, int translate_link) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 v = in[i];
    if (translate_link) { const uint32_t l = __float_as_uint(v.w); if (l < n) v.w = __uint_as_float(pid[l]); }
    out[pid[i]] = v;
}
__global__ __launch_bounds__(256) void k_translate_heads(size_t cells, uint32_t n, const uint32_t* __restrict__ pid, uint32_t* __restrict__ heads) {
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < cells; c += (size_t)gridDim.x * 256) {
        const uint32_t hd = heads[c];
        if (hd != 0u && hd - 1u < n) heads[c] = pid[hd - 1u] + 1u;
    }
}
__global__ __launch_bounds__(256) void k_iota(uint32_t first, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[first + i] = first + i;
}

// ---- static marker pattern: transfer_clear.comp:10-14 + transfer_set_boundary_marker.comp:11-19 --------------------
__device__ __forceinline__ uint32_t static_marker_quad(const Grid& g, const float4* __restrict__ solid, int base, int x0, int y, int z) {
    const bool shell_yz = (y == 0) | (z == 0) | (y == g.ny - 1) | (z == g.nz - 1);
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = x0 + j;
        bool sol = shell_yz | (x == 0) | (x == g.nx - 1);
        if (!sol && solid) sol = solid[base + j].w != 0.0f;
        packed |= (sol ? 0u : 0xFFu) << (8 * j);
    }
    return packed;
}
// T1 (+T3) over the reset list: marker := static pattern, list heads := 0; stale bricks additionally get velocity and
// pressure volumes zeroed (what the reference's dense D2 / pressure_init / R2 passes would have written there).
__global__ __launch_bounds__(BRICK_THREADS) void k_reset_bricks(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                const float4* __restrict__ solid, int8_t* __restrict__ marker,
                                                                uint32_t* __restrict__ ll0, uint32_t* __restrict__ ll1, uint32_t* __restrict__ ll2,
                                                                float* __restrict__ vx, float* __restrict__ vy, float* __restrict__ vz,
                                                                float* __restrict__ p0, float* __restrict__ p1) {
    const uint32_t n = *count;
    for (uint32_t j = blockIdx.x; j < list_slots(n); j += gridDim.x) {
        uint32_t i;
        if (!list_slot(j, n, i)) continue;
        const uint32_t e = list[i];
        int x0, y, z;
        if (!brick_quad(bg, e & ~STALE_BIT, threadIdx.x, x0, y, z)) continue;
        const int base = cidx(bg.g, x0, y, z);
        *reinterpret_cast<uint32_t*>(marker + base) = static_marker_quad(bg.g, solid, base, x0, y, z);
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        if (ll0) *reinterpret_cast<uint4*>(ll0 + base) = z4;
        if (ll1) *reinterpret_cast<uint4*>(ll1 + base) = z4;
        if (ll2) *reinterpret_cast<uint4*>(ll2 + base) = z4;
        if ((e & STALE_BIT) && vx) {
            const float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(vx + base) = f0; *reinterpret_cast<float4*>(vy + base) = f0; *reinterpret_cast<float4*>(vz + base) = f0;
            *reinterpret_cast<float4*>(p0 + base) = f0; *reinterpret_cast<float4*>(p1 + base) = f0;
        }
    }
}
// Dense variant (creation, new solid voxels): every cell gets the static pattern.
__global__ __launch_bounds__(256) void k_static_marker_dense(Grid g, const float4* __restrict__ solid, int8_t* __restrict__ marker, int z_lo, int z_hi) {
    const int qpp = (g.nx >> 2) * g.ny, nquads = qpp * (z_hi - z_lo);      // (the planes [z_lo, z_hi) this domain's volumes hold)
    for (int q0 = blockIdx.x * 256 + threadIdx.x; q0 < nquads; q0 += gridDim.x * 256) {
        const int base = (q0 + qpp * z_lo) << 2;
        const int x0 = base % g.nx, yz = base / g.nx, y = yz % g.ny, z = yz / g.ny;
        *reinterpret_cast<uint32_t*>(marker + base) = static_marker_quad(g, solid, base, x0, y, z);
    }
}

// ---- T4 (transfer_gather_velocity.comp:39-127) and R1 (density_projection_gather_error.comp:41-198) over brick lists --------
// One workgroup per brick: (16+1)x(8+1)x(4+1) = 765 list cells (brick + one halo layer on the negative sides), 768 threads = 12 full waves.
constexpr int GT_X = BX + 1, GT_Y = BY + 1, GT_Z = BZ + 1, GT_N = GT_X * GT_Y * GT_Z;   // 17 x 9 x 5 = 765

// LDS-only workgroup barrier: waits for this wave's LDS operations, NOT for its outstanding global loads, so loads requested ahead
// stay in flight across the exchange (a __syncthreads() would drain vmcnt first).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct GatherArgs3 { const uint32_t* heads[3]; float* out[3]; float gravity_dt[3]; uint32_t node_stride; };

// "Partial sum" formulation.  (Round 1 moved PARTICLES to faces -- per round every thread published one particle through LDS and read
// seven, with a 12-wave barrier per round: measured LDS-issue / barrier bound; removed.)  Every
// thread walks its OWN dual cell's list once (<= 12 / 32 nodes, transfer_gather_velocity.comp:61, density_projection_gather_error
// .comp:69 -- the caps are per list, so the same particles take part) and accumulates, in registers, that list's contribution to the
// EIGHT samples it reaches (dual cell d feeds the faces d + {0,1}^3); the partial sums are exchanged through LDS ONCE per brick and
// each face adds up the eight partials of the eight lists the reference walks for it.  Per particle the eight weights share their
// factors (2 x 3 one-dimensional hats), which halves the arithmetic; every product / sum of a particle-face pair is formed exactly
// as in add_particle(), only the ORDER in which a face's contributions are added differs (list-major instead of round-major).
constexpr int GATHER_CAP_V = 12;      // transfer_gather_velocity.comp:61
constexpr int GP_STRIDE = 768;
struct GatherPartialsV { float2 part[8][GP_STRIDE]; };     // [corner][list cell] {sum w*d, sum w}: 48 KiB
struct GatherPartialsD { float part[8][GP_STRIDE]; };      // [corner][list cell] sum w: 24 KiB

template <int COMP>
__device__ __forceinline__ void gather_walk(const GatherNode* __restrict__ nodes, uint32_t cur, int gx, int gy, int gz, float (&v)[8], float (&ws)[8]) {      // nodes: component COMP's array
    // the two sample coordinates per axis this list reaches: faces d and d + 1 (:20, sample = face + 0.5 (+ 0.5 along COMP))
    const float sx0 = (float)gx + 0.5f + (COMP == 0 ? 0.5f : 0.0f), sx1 = (float)(gx + 1) + 0.5f + (COMP == 0 ? 0.5f : 0.0f);
    const float sy0 = (float)gy + 0.5f + (COMP == 1 ? 0.5f : 0.0f), sy1 = (float)(gy + 1) + 0.5f + (COMP == 1 ? 0.5f : 0.0f);
    const float sz0 = (float)gz + 0.5f + (COMP == 2 ? 0.5f : 0.0f), sz1 = (float)(gz + 1) + 0.5f + (COMP == 2 ? 0.5f : 0.0f);
    // one hop = the two halves of ONE 32-byte node (k_build_lists): {position, link}, {row}
    const float4* nd = reinterpret_cast<const float4*>(nodes);
    float4 p = nd[2 * (size_t)cur], r = nd[2 * (size_t)cur + 1];
    uint32_t nxt = __float_as_uint(p.w);
    for (int round = 0; round < GATHER_CAP_V; ++round) {                                           // :61
        const bool has_n = nxt != INVALID_LL && round + 1 < GATHER_CAP_V;
        float4 pn = p, rn = r; uint32_t nn = INVALID_LL;
        if (has_n) { pn = nd[2 * (size_t)nxt]; rn = nd[2 * (size_t)nxt + 1]; nn = __float_as_uint(pn.w); }   // next node in flight during the arithmetic
        const float tx[2] = {sx0 - p.x, sx1 - p.x}, ty[2] = {sy0 - p.y, sy1 - p.y}, tz[2] = {sz0 - p.z, sz1 - p.z};   // :20
        const float ox[2] = {satf(1.0f - fabsf(tx[0])), satf(1.0f - fabsf(tx[1]))};
        const float oy[2] = {satf(1.0f - fabsf(ty[0])), satf(1.0f - fabsf(ty[1]))};
        const float oz[2] = {satf(1.0f - fabsf(tz[0])), satf(1.0f - fabsf(tz[1]))};
        const float ax[2] = {r.x * tx[0], r.x * tx[1]}, ay[2] = {r.y * ty[0], r.y * ty[1]}, az[2] = {r.z * tz[0], r.z * tz[1]};
        const float rw = r.w * 1.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kx = k & 1, ky = (k >> 1) & 1, kz = k >> 2;
            const float w = ox[kx] * oy[ky] * oz[kz];                                    // :22
            const float d = ((ax[kx] + ay[ky]) + az[kz]) + rw;                           // :24
            v[k] += w * d;
            ws[k] += w;
        }
        if (!has_n) break;
        p = pn; r = rn; nxt = nn;
    }
}

template <int COMP>
__device__ __forceinline__ void gather_velocity_partial_body(GatherPartialsV& sh, uint32_t first_brick, uint32_t brick_stride, const BrickGeom& bg, const uint32_t* __restrict__ list,
                                                             const uint32_t* __restrict__ count, const int8_t* __restrict__ marker, const uint32_t* __restrict__ heads,
                                                             const GatherNode* __restrict__ nodes, float* __restrict__ out, float gravity_dt) {
    const Grid g = bg.g;
    const int tid = threadIdx.x;
    const bool live = tid < GT_N;
    const int lx = tid % GT_X, ly = (tid / GT_X) % GT_Y, lz = tid / (GT_X * GT_Y);
    const uint32_t n = *count;
    for (uint32_t i = first_brick; i < n; i += brick_stride) {
        const uint32_t b = list[i];
        int bx, by, bz; brick_coords(bg, b, bx, by, bz);
        const int gx = bx * BX + lx - 1, gy = by * BY + ly - 1, gz = bz * BZ + lz - 1;      // :41
        const bool in = live && inb(g, gx, gy, gz);
        uint32_t cur = in ? heads[cidx(g, gx, gy, gz)] - 1u : INVALID_LL;
        bool has = cur != INVALID_LL;
        // (the barrier also separates the previous brick's reads of the partials from this brick's writes)
        if (!__syncthreads_or(has)) continue;      // no particle anywhere in the tile: no face of this brick touches a FLUID cell, nothing is written
        float v[8], ws[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = 0.0f; ws[k] = 0.0f; }
        if (has) gather_walk<COMP>(nodes, cur, gx, gy, gz, v, ws);
        if (live) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sh.part[k][tid] = make_float2(v[k], ws[k]);
        }
        __syncthreads();
        const bool border = !live || lx == 0 || ly == 0 || lz == 0;
        if (!border && in) {
            const int mA = (int)marker[cidx(g, gx, gy, gz)];
            const int mB = mk(marker, g, gx + (COMP == 0), gy + (COMP == 1), gz + (COMP == 2));
            if (mA == CELL_FLUID || mB == CELL_FLUID) {                                          // :50
                float val = 0.0f;
                if (mA != CELL_SOLID && mB != CELL_SOLID) {                                      // :51
                    float wsum = 0.0f;
                    // the face's eight lists in the reference's order (:87-93): own cell, then the seven negative neighbours
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int kx = k & 1, ky = (k >> 1) & 1, kz = k >> 2;
                        const float2 q = sh.part[k][tid - kx - ky * GT_X - kz * GT_X * GT_Y];
                        val += q.x; wsum += q.y;
                    }
                    if (wsum > 0.0f) val /= wsum;                                                // :117-119
                    val += gravity_dt;                                                           // :120
                }
                out[cidx(g, gx, gy, gz)] = val;                                                  // (:121-124: 0 with exactly one solid side)
            }
        }
    }
}
// Work mapping: the three component gathers of a brick read the SAME particle positions.  Block b runs on XCD b % 8 (observed dispatch
// order, a speed hint only); consecutive blocks of one XCD take the three components of one brick slot, so the positions fetched for the
// first component are L2 hits for the other two.  gridDim.x = 3 * slots, slots a multiple of 8.
__device__ __forceinline__ void gather3_block_role(uint32_t& slot, uint32_t& slots, int& comp) {
    const uint32_t b = blockIdx.x, xcd = b & 7u, s = b >> 3;
    comp = (int)(s % 3u);
    slot = (s / 3u) * 8u + xcd;
    slots = gridDim.x / 3u;
}
__global__ __launch_bounds__(768) void k_gather_velocity3_p(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                            const int8_t* __restrict__ marker, const GatherNode* __restrict__ nodes, GatherArgs3 a) {
    __shared__ GatherPartialsV sh;
    uint32_t slot, slots; int comp;
    gather3_block_role(slot, slots, comp);
    if (comp == 0) gather_velocity_partial_body<0>(sh, slot, slots, bg, list, count, marker, a.heads[0], nodes + (size_t)0 * a.node_stride, a.out[0], a.gravity_dt[0]);
    else if (comp == 1) gather_velocity_partial_body<1>(sh, slot, slots, bg, list, count, marker, a.heads[1], nodes + (size_t)1 * a.node_stride, a.out[1], a.gravity_dt[1]);
    else gather_velocity_partial_body<2>(sh, slot, slots, bg, list, count, marker, a.heads[2], nodes + (size_t)2 * a.node_stride, a.out[2], a.gravity_dt[2]);
}

// ---- the same gather with the tile's NON-EMPTY lists compacted -------------------------------------------------------------------
// The kernel above gives every list cell of the tile a lane; a brick of the headline scene holds particles in a fifth of its cells, so
// four lanes of five only wait at the exchange barrier, and with 12 waves per workgroup two workgroups fill a CU.  Here 256 threads load
// the 765 heads, compact the non-empty lists IN CELL ORDER (ballots + a 12-entry scan), and thread t walks the list in slot t: every
// lane of the walk phase is busy and 6-8 workgroups fit a CU.  Tiles with more than 256 lists take several passes, last slots first:
// a face's eight lists d - {0,1}^3 come in DESCENDING cell order for k = 0..7, so a running sum over (pass descending, k ascending)
// adds the eight partial sums in exactly the reference's list order (:87-93) -- bit-identical to k_gather_velocity3_p.
constexpr int GS_THREADS = 256;
constexpr int GS_CHUNKS = 3 * (GS_THREADS / 64);      // (pass over the tile, wave): 64 consecutive tile cells each
constexpr uint32_t GS_EMPTY = 0xFFFFu;
struct GatherSparseShared {
    float2 part[8][GS_THREADS];      // [corner][slot of this pass] {sum w*d, sum w}: 16 KiB
    uint16_t cell_of[GT_N + 3];      // slot -> tile cell
    uint16_t slot_of[GT_N + 3];      // tile cell -> slot (GS_EMPTY: no list)
    uint32_t chunk[2][GS_CHUNKS];    // non-empty lists per chunk; double buffered by brick parity
};

template <int COMP>
__device__ __forceinline__ void gather_velocity_sparse_body(GatherSparseShared& sh, uint32_t first_brick, uint32_t brick_stride, const BrickGeom& bg,
                                                            const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, const int8_t* __restrict__ marker,
                                                            const uint32_t* __restrict__ heads, const GatherNode* __restrict__ nodes, float* __restrict__ out, float gravity_dt) {
    const Grid g = bg.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = *count;
    int parity = 0;
    for (uint32_t i = first_brick; i < n; i += brick_stride, parity ^= 1) {
        const uint32_t b = list[i];
        int bx, by, bz; brick_coords(bg, b, bx, by, bz);
        const int ox0 = bx * BX - 1, oy0 = by * BY - 1, oz0 = bz * BZ - 1;                   // grid coordinates of tile cell 0 (:41)
        // ---- compact the tile's non-empty lists, in cell order ----
        unsigned long long mask[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = tid + GS_THREADS * j;
            const int gx = ox0 + c % GT_X, gy = oy0 + (c / GT_X) % GT_Y, gz = oz0 + c / (GT_X * GT_Y);
            mask[j] = __ballot(c < GT_N && inb(g, gx, gy, gz) && heads[cidx(g, gx, gy, gz)] != 0u);
            if (lane == 0) sh.chunk[parity][j * (GS_THREADS / 64) + wave] = (uint32_t)__popcll(mask[j]);
        }
        __syncthreads();                // (the chunk counts of the brick before the last were read before this barrier: the other buffer is free)
        uint32_t nl = 0, before[3] = {0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < GS_CHUNKS; ++ch) {
            const uint32_t v = sh.chunk[parity][ch];
#pragma unroll
            for (int j = 0; j < 3; ++j) if (ch < j * (GS_THREADS / 64) + wave) before[j] += v;
            nl += v;
        }
        if (nl == 0) continue;          // no particle anywhere in the tile: no face of this brick touches a FLUID cell, nothing is written
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = tid + GS_THREADS * j;
            const bool has = (mask[j] >> lane) & 1ull;
            const uint32_t slot = before[j] + (uint32_t)__popcll(mask[j] & ((1ull << lane) - 1ull));
            if (has) sh.cell_of[slot] = (uint16_t)c;
            if (c < GT_N) sh.slot_of[c] = has ? (uint16_t)slot : (uint16_t)GS_EMPTY;
        }
        __syncthreads();
        // ---- walk: thread t takes the list in slot pass * 256 + t; the faces add up what they reach ----
        float val[2] = {0.0f, 0.0f}, wsum[2] = {0.0f, 0.0f};
        for (int pass = (int)((nl - 1u) / GS_THREADS); pass >= 0; --pass) {
            const uint32_t s = (uint32_t)pass * GS_THREADS + tid;
            if (s < nl) {
                const int c = sh.cell_of[s];
                const int gx = ox0 + c % GT_X, gy = oy0 + (c / GT_X) % GT_Y, gz = oz0 + c / (GT_X * GT_Y);
                float v[8], ws[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[k] = 0.0f; ws[k] = 0.0f; }
                gather_walk<COMP>(nodes, heads[cidx(g, gx, gy, gz)] - 1u, gx, gy, gz, v, ws);
#pragma unroll
                for (int k = 0; k < 8; ++k) sh.part[k][tid] = make_float2(v[k], ws[k]);
            }
            __syncthreads();
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int fc = tid + GS_THREADS * f;                                        // face (fc % 16, fc / 16 % 8, fc / 128) of the brick
                const int tc = ((fc & 15) + 1) + (((fc >> 4) & 7) + 1) * GT_X + ((fc >> 7) + 1) * GT_X * GT_Y;
                // the face's eight lists in the reference's order (:87-93): own cell, then the seven negative neighbours
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t sl = sh.slot_of[tc - (k & 1) - ((k >> 1) & 1) * GT_X - (k >> 2) * GT_X * GT_Y];
                    if (sl != GS_EMPTY && (int)(sl / GS_THREADS) == pass) { const float2 q = sh.part[k][sl % GS_THREADS]; val[f] += q.x; wsum[f] += q.y; }
                }
            }
            __syncthreads();            // (also: the next brick's compaction overwrites slot_of / cell_of)
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int fc = tid + GS_THREADS * f;
            const int gx = bx * BX + (fc & 15), gy = by * BY + ((fc >> 4) & 7), gz = bz * BZ + (fc >> 7);
            if (!inb(g, gx, gy, gz)) continue;
            const int mA = (int)marker[cidx(g, gx, gy, gz)];
            const int mB = mk(marker, g, gx + (COMP == 0), gy + (COMP == 1), gz + (COMP == 2));
            if (mA == CELL_FLUID || mB == CELL_FLUID) {                                          // :50
                float o = 0.0f;
                if (mA != CELL_SOLID && mB != CELL_SOLID) {                                      // :51
                    o = val[f];
                    if (wsum[f] > 0.0f) o /= wsum[f];                                            // :117-119
                    o += gravity_dt;                                                             // :120
                }
                out[cidx(g, gx, gy, gz)] = o;                                                    // (:121-124: 0 with exactly one solid side)
            }
        }
    }
}
__global__ __launch_bounds__(GS_THREADS) void k_gather_velocity3_s(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                   const int8_t* __restrict__ marker, const GatherNode* __restrict__ nodes, GatherArgs3 a) {
    __shared__ GatherSparseShared sh;
    uint32_t slot, slots; int comp;
    gather3_block_role(slot, slots, comp);
    if (comp == 0) gather_velocity_sparse_body<0>(sh, slot, slots, bg, list, count, marker, a.heads[0], nodes + (size_t)0 * a.node_stride, a.out[0], a.gravity_dt[0]);
    else if (comp == 1) gather_velocity_sparse_body<1>(sh, slot, slots, bg, list, count, marker, a.heads[1], nodes + (size_t)1 * a.node_stride, a.out[1], a.gravity_dt[1]);
    else gather_velocity_sparse_body<2>(sh, slot, slots, bg, list, count, marker, a.heads[2], nodes + (size_t)2 * a.node_stride, a.out[2], a.gravity_dt[2]);
}

// ---- T4, list-centric (round 6; single domains, well-filled bricks): every list is walked ONCE ----------------------------------------------------------------
// The kernels above give every brick the lists of its tile (own 16 x 8 x 4 cells + the negative halo, 765 cells): a halo list is walked by
// two to eight bricks, 1.49 x the hops, and a hop -- two divergent 16-byte accesses per lane -- is what the walk is bound by (one L1 -> L2
// request per hop at ~7 cycles per request and CU; profiles/r06_pmc_dam_halfhalf_highres_*.csv).  Here a brick walks its OWN 512 lists only.
// Their partial sums reach the 17 x 9 x 5 faces own cell + {0,1}^3 (the REGION: the brick's own faces and one layer on the POSITIVE sides);
// per region face the kernel adds up the partials it holds, in the reference's list order (:87-93).  Faces whose eight lists all lie in
// this brick (region coordinates >= 1 and inside the brick: 315 of 512) are finished here; the sums of the others -- own faces that also
// take part in a negative neighbour's lists, and the positive layer, which belongs to the neighbours -- go to a scratch array, stamped with
// the sequence number of this step's list build, and k_gather_finish3 adds up to eight of them per boundary face: own brick first, then
// the negative neighbours in the order of their lists.  Only the ASSOCIATION of a boundary face's sum changes (per-brick subtotals).
struct GatherHalo { float2* sums; uint32_t* stamp; uint32_t seq, nb; };      // sums[(c nb + brick) 765 + region face], stamp[c nb + brick] = seq: the brick wrote sums this step
struct GatherOwnShared { float2 part[8][BX * BY * BZ]; };                      // [corner][own list cell]: 32 KiB
__device__ __forceinline__ float2 region_sum(const float2 (&part)[8][BX * BY * BZ], int fx, int fy, int fz) {
    float2 S = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int lx = fx - (k & 1), ly = fy - ((k >> 1) & 1), lz = fz - (k >> 2);
        if ((unsigned)lx < (unsigned)BX && (unsigned)ly < (unsigned)BY && (unsigned)lz < (unsigned)BZ) { const float2 q = part[k][lx + BX * (ly + BY * lz)]; S.x += q.x; S.y += q.y; }
    }
    return S;
}
__device__ __forceinline__ bool region_interior(int fx, int fy, int fz) { return fx >= 1 && fx < BX && fy >= 1 && fy < BY && fz >= 1 && fz < BZ; }
// the value of a face from its sums (transfer_gather_velocity.comp:50-51, 117-124)
template <int COMP>
__device__ __forceinline__ void gather_finish_face(const Grid& g, const int8_t* __restrict__ marker, float* __restrict__ out, float gravity_dt, int gx, int gy, int gz, float2 S) {
    if (!inb(g, gx, gy, gz)) return;
    const int mA = (int)marker[cidx(g, gx, gy, gz)];
    const int mB = mk(marker, g, gx + (COMP == 0), gy + (COMP == 1), gz + (COMP == 2));
    if (mA == CELL_FLUID || mB == CELL_FLUID) {                                                  // :50
        float o = 0.0f;
        if (mA != CELL_SOLID && mB != CELL_SOLID) {                                              // :51
            o = S.x;
            if (S.y > 0.0f) o /= S.y;                                                            // :117-119
            o += gravity_dt;                                                                     // :120
        }
        out[cidx(g, gx, gy, gz)] = o;                                                            // (:121-124: 0 with exactly one solid side)
    }
}
template <int COMP>
__device__ __forceinline__ void gather_region_out(const GatherOwnShared& sh, const BrickGeom& bg, uint32_t b, int bx, int by, int bz, const int8_t* __restrict__ marker,
                                                  float* __restrict__ out, float gravity_dt, float2* __restrict__ sums, int nthreads) {
    for (int r = threadIdx.x; r < GT_N; r += nthreads) {
        const int fx = r % GT_X, fy = (r / GT_X) % GT_Y, fz = r / (GT_X * GT_Y);
        const float2 S = region_sum(sh.part, fx, fy, fz);
        if (region_interior(fx, fy, fz)) gather_finish_face<COMP>(bg.g, marker, out, gravity_dt, bx * BX + fx, by * BY + fy, bz * BZ + fz, S);
        else sums[(size_t)b * GT_N + r] = S;
    }
}
// one lane per own list cell (512 threads): the counterpart of k_gather_velocity3_p
template <int COMP>
__device__ __forceinline__ void gather_velocity_own_body(GatherOwnShared& sh, uint32_t first_brick, uint32_t brick_stride, const BrickGeom& bg, const uint32_t* __restrict__ list,
                                                         const uint32_t* __restrict__ count, const int8_t* __restrict__ marker, const uint32_t* __restrict__ heads,
                                                         const GatherNode* __restrict__ nodes, float* __restrict__ out, float gravity_dt, float2* __restrict__ sums,
                                                         uint32_t* __restrict__ stamp, uint32_t seq) {
    const Grid g = bg.g;
    const int tid = threadIdx.x, lx = tid % BX, ly = (tid / BX) % BY, lz = tid / (BX * BY);
    const uint32_t n = *count;
    for (uint32_t i = first_brick; i < n; i += brick_stride) {
        const uint32_t b = list[i];
        int bx, by, bz; brick_coords(bg, b, bx, by, bz);
        const int gx = bx * BX + lx, gy = by * BY + ly, gz = bz * BZ + lz;
        const uint32_t cur = inb(g, gx, gy, gz) ? heads[cidx(g, gx, gy, gz)] - 1u : INVALID_LL;
        // (the barrier also separates the previous brick's reads of the partials from this brick's writes)
        if (!__syncthreads_or(cur != INVALID_LL)) continue;      // none of the brick's own lists holds a particle: it has nothing to add to any face
        float v[8], ws[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = 0.0f; ws[k] = 0.0f; }
        if (cur != INVALID_LL) gather_walk<COMP>(nodes, cur, gx, gy, gz, v, ws);
#pragma unroll
        for (int k = 0; k < 8; ++k) sh.part[k][tid] = make_float2(v[k], ws[k]);
        if (tid == 0) stamp[b] = seq;
        __syncthreads();
        gather_region_out<COMP>(sh, bg, b, bx, by, bz, marker, out, gravity_dt, sums, BX * BY * BZ);
    }
}
struct GatherOwnArgs3 { const uint32_t* heads[3]; float* out[3]; float gravity_dt[3]; uint32_t node_stride; GatherHalo halo; };
__global__ __launch_bounds__(BX * BY * BZ) void k_gather_velocity3_po(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                      const int8_t* __restrict__ marker, const GatherNode* __restrict__ nodes, GatherOwnArgs3 a) {
    __shared__ GatherOwnShared sh;
    uint32_t slot, slots; int comp;
    gather3_block_role(slot, slots, comp);
    float2* const sums = a.halo.sums + (size_t)comp * a.halo.nb * GT_N;
    uint32_t* const stamp = a.halo.stamp + (size_t)comp * a.halo.nb;
    if (comp == 0) gather_velocity_own_body<0>(sh, slot, slots, bg, list, count, marker, a.heads[0], nodes + (size_t)0 * a.node_stride, a.out[0], a.gravity_dt[0], sums, stamp, a.halo.seq);
    else if (comp == 1) gather_velocity_own_body<1>(sh, slot, slots, bg, list, count, marker, a.heads[1], nodes + (size_t)1 * a.node_stride, a.out[1], a.gravity_dt[1], sums, stamp, a.halo.seq);
    else gather_velocity_own_body<2>(sh, slot, slots, bg, list, count, marker, a.heads[2], nodes + (size_t)2 * a.node_stride, a.out[2], a.gravity_dt[2], sums, stamp, a.halo.seq);
}
// The faces on the three negative sides of a brick (197 of 512): own sums + those of the (up to seven) negative neighbours whose positive layer they lie in.
template <int COMP>
__device__ __forceinline__ void gather_finish_body(uint32_t first_brick, uint32_t brick_stride, const BrickGeom& bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                   const int8_t* __restrict__ marker, float* __restrict__ out, float gravity_dt, const float2* __restrict__ sums,
                                                   const uint32_t* __restrict__ stamp, uint32_t seq) {
    const uint32_t n = *count;
    for (uint32_t i = first_brick; i < n; i += brick_stride) {
        const uint32_t b = list[i];
        int bx, by, bz; brick_coords(bg, b, bx, by, bz);
        uint32_t live = 0;              // bit d: brick - d (d in {0,1}^3) exists and wrote sums this step
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int qx = bx - (d & 1), qy = by - ((d >> 1) & 1), qz = bz - (d >> 2);
            if (qx >= 0 && qy >= 0 && qz >= 0 && stamp[(qz * bg.nby + qy) * bg.nbx + qx] == seq) live |= 1u << d;
        }
        if (!live) continue;            // (uniform) no particle near any face of this brick
        for (int f = threadIdx.x; f < BX * BY * BZ; f += blockDim.x) {
            const int fx = f % BX, fy = (f / BX) % BY, fz = f / (BX * BY);
            if (region_interior(fx, fy, fz)) continue;
            float2 S = make_float2(0.0f, 0.0f);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int dx = d & 1, dy = (d >> 1) & 1, dz = d >> 2;
                if (!((live >> d) & 1u) || (dx && fx) || (dy && fy) || (dz && fz)) continue;
                const uint32_t q = (uint32_t)(((bz - dz) * bg.nby + (by - dy)) * bg.nbx + (bx - dx));
                const float2 t = sums[(size_t)q * GT_N + (fx + dx * BX) + GT_X * ((fy + dy * BY) + GT_Y * (fz + dz * BZ))];
                S.x += t.x; S.y += t.y;
            }
            gather_finish_face<COMP>(bg.g, marker, out, gravity_dt, bx * BX + fx, by * BY + fy, bz * BZ + fz, S);
        }
    }
}
__global__ __launch_bounds__(256) void k_gather_finish3(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, const int8_t* __restrict__ marker, GatherOwnArgs3 a) {
    uint32_t slot, slots; int comp;
    gather3_block_role(slot, slots, comp);
    const float2* const sums = a.halo.sums + (size_t)comp * a.halo.nb * GT_N;
    const uint32_t* const stamp = a.halo.stamp + (size_t)comp * a.halo.nb;
    if (comp == 0) gather_finish_body<0>(slot, slots, bg, list, count, marker, a.out[0], a.gravity_dt[0], sums, stamp, a.halo.seq);
    else if (comp == 1) gather_finish_body<1>(slot, slots, bg, list, count, marker, a.out[1], a.gravity_dt[1], sums, stamp, a.halo.seq);
    else gather_finish_body<2>(slot, slots, bg, list, count, marker, a.out[2], a.gravity_dt[2], sums, stamp, a.halo.seq);
}

// R1 in the same formulation (density_projection_gather_error.comp:41-198): samples are cell centres, the list cap is 32
__global__ __launch_bounds__(768) void k_density_gather_p(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                          const int8_t* __restrict__ marker, const uint32_t* __restrict__ heads,
                                                          const float4* __restrict__ pos, float* __restrict__ residual, float dt) {
    __shared__ GatherPartialsD sh;
    const Grid g = bg.g;
    const int tid = threadIdx.x;
    const bool live = tid < GT_N;
    const int lx = tid % GT_X, ly = (tid / GT_X) % GT_Y, lz = tid / (GT_X * GT_Y);
    const uint32_t n = *count;
    for (uint32_t j = blockIdx.x; j < list_slots(n); j += gridDim.x) {
        uint32_t i;
        if (!list_slot(j, n, i)) continue;
        const uint32_t b = list[i];
        int bx, by, bz; brick_coords(bg, b, bx, by, bz);
        const int gx = bx * BX + lx - 1, gy = by * BY + ly - 1, gz = bz * BZ + lz - 1;
        const bool in = live && inb(g, gx, gy, gz);
        uint32_t cur = in ? heads[cidx(g, gx, gy, gz)] - 1u : INVALID_LL;
        const bool has = cur != INVALID_LL;
        __syncthreads();          // the previous brick's reads of the partials are done (a FLUID brick always holds particles: no early-out)
        float ws[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ws[k] = 0.0f;
        if (has) {
            const float sx0 = (float)gx + 0.5f, sx1 = (float)(gx + 1) + 0.5f, sy0 = (float)gy + 0.5f, sy1 = (float)(gy + 1) + 0.5f, sz0 = (float)gz + 0.5f, sz1 = (float)(gz + 1) + 0.5f;
            float4 p = pos[cur];
            for (int round = 0; round < 32; ++round) {                                            // :69
                const uint32_t nxt = __float_as_uint(p.w);
                const bool has_n = nxt != INVALID_LL && round + 1 < 32;
                float4 pn = p;
                if (has_n) pn = pos[nxt];
                const float ox[2] = {satf(1.0f - fabsf(sx0 - p.x)), satf(1.0f - fabsf(sx1 - p.x))};
                const float oy[2] = {satf(1.0f - fabsf(sy0 - p.y)), satf(1.0f - fabsf(sy1 - p.y))};
                const float oz[2] = {satf(1.0f - fabsf(sz0 - p.z)), satf(1.0f - fabsf(sz1 - p.z))};
#pragma unroll
                for (int k = 0; k < 8; ++k) ws[k] += ox[k & 1] * oy[(k >> 1) & 1] * oz[k >> 2];   // :27-31
                if (!has_n) break;
                p = pn;
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sh.part[k][tid] = ws[k];
        }
        __syncthreads();
        const bool border = !live || lx == 0 || ly == 0 || lz == 0;
        if (!border && in && marker[cidx(g, gx, gy, gz)] == CELL_FLUID) {                         // :46
            float density = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) density += sh.part[k][tid - (k & 1) - ((k >> 1) & 1) * GT_X - (k >> 2) * GT_X * GT_Y];
            const int m[6] = {mk(marker, g, gx + 1, gy, gz), mk(marker, g, gx, gy + 1, gz), mk(marker, g, gx, gy, gz + 1),
                              mk(marker, g, gx - 1, gy, gz), mk(marker, g, gx, gy - 1, gz), mk(marker, g, gx, gy, gz - 1)};   // :115-120
            bool anyAir = false;
#pragma unroll
            for (int k = 0; k < 6; ++k) { if (m[k] == CELL_SOLID) density += 0.5625f; if (m[k] == CELL_AIR) anyAir = true; }   // :167-179
            if (anyAir) density = fmaxf(8.0f, density);                                           // :182-184
            density = 1.0f - density / 8.0f;                                                      // :188
            density = clampf(density, -0.5f, 0.5f);                                               // :192
            density /= dt;                                                                        // :196
            residual[cidx(g, gx, gy, gz)] = density;
        }
    }
}

// ---- quad-vectorised element-wise grid kernels over brick lists -------------------------------------------------------
#define BRICK_LOOP_BEGIN(bg, list, count)                                                     \
    const uint32_t _n = *(count);                                                             \
    for (uint32_t _j = blockIdx.x; _j < list_slots(_n); _j += gridDim.x) {                    \
        uint32_t _i;                                                                          \
        if (!list_slot(_j, _n, _i)) continue;                                                 \
        int x0, y, z;                                                                         \
        if (!brick_quad((bg), (list)[_i] & ~STALE_BIT, threadIdx.x, x0, y, z)) continue;      \
        const int base = cidx((bg).g, x0, y, z);
#define BRICK_LOOP_END }

__device__ __forceinline__ float solid_comp(const float4* __restrict__ solid, const Grid& g, int x, int y, int z, int comp) {
    return (solid && inb(g, x, y, z)) ? comp3(solid[cidx(g, x, y, z)], comp) : 0.0f;
}

// D1: divergence_compute.comp:28-87, one quad: the FLUID lanes of rr become the divergence, the others keep what they hold.  `m`: the quad's markers and
// its six neighbours' (load_quad_markers).  Shared by k_divergence_b and by the init kernel of the velocity solve, which forms b = div u on the fly
// inside a step (k_pcg_init_b<true>: one launch and one pass over the residual volume less).
struct DivergenceSrc { const float* vx; const float* vy; const float* vz; const float4* solid; };
__device__ __forceinline__ void divergence_quad(const Grid& g, const QuadMarkers& m, const DivergenceSrc& v, int base, int x0, int y, int z, float (&rr)[4]) {
    const int plane = g.nx * g.ny;
    const float4 px = ld4(v.vx + base), py = ld4(v.vy + base), pz = ld4(v.vz + base);
    const float qx_edge = x0 > 0 ? v.vx[base - 1] : 0.0f;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 qy = y > 0 ? ld4(v.vy + base - g.nx) : zero;
    const float4 qz = z > 0 ? ld4(v.vz + base - plane) : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (mbyte(m.c, j) != CELL_FLUID) continue;
        const int x = x0 + j;
        const float vpx = f4(px, j), vpy = f4(py, j), vpz = f4(pz, j);
        const float vqx = j > 0 ? f4(px, j - 1) : qx_edge, vqy = f4(qy, j), vqz = f4(qz, j);
        float div = vpx - vqx;
        div += vpy - vqy;
        div += vpz - vqz;
        const int mX0 = j > 0 ? mbyte(m.c, j - 1) : m.xm, mX1 = j < 3 ? mbyte(m.c, j + 1) : m.xp;
        div += (mX0 == CELL_SOLID) ? vqx - solid_comp(v.solid, g, x - 1, y, z, 0) : 0.0f;       // :67-75
        div += (mbyte(m.ym, j) == CELL_SOLID) ? vqy - solid_comp(v.solid, g, x, y - 1, z, 1) : 0.0f;
        div += (mbyte(m.zm, j) == CELL_SOLID) ? vqz - solid_comp(v.solid, g, x, y, z - 1, 2) : 0.0f;
        div -= (mX1 == CELL_SOLID) ? vpx - solid_comp(v.solid, g, x + 1, y, z, 0) : 0.0f;       // :76-84
        div -= (mbyte(m.yp, j) == CELL_SOLID) ? vpy - solid_comp(v.solid, g, x, y + 1, z, 1) : 0.0f;
        div -= (mbyte(m.zp, j) == CELL_SOLID) ? vpz - solid_comp(v.solid, g, x, y, z + 1, 2) : 0.0f;
        rr[j] = div;
    }
}
__global__ __launch_bounds__(BRICK_THREADS) void k_divergence_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                const int8_t* __restrict__ marker, DivergenceSrc v, float* __restrict__ residual) {
    const Grid g = bg.g;
    BRICK_LOOP_BEGIN(bg, list, count)
        QuadMarkers m; m.c = *reinterpret_cast<const uint32_t*>(marker + base);
        if (!any_fluid4(m.c)) continue;
        load_quad_markers(marker, g, base, x0, y, z, m);
        const float4 rc = ld4(residual + base);
        float rr[4] = {rc.x, rc.y, rc.z, rc.w};
        divergence_quad(g, m, v, base, x0, y, z, rr);
        *reinterpret_cast<float4*>(residual + base) = make_float4(rr[0], rr[1], rr[2], rr[3]);
    BRICK_LOOP_END
}

// D2: divergence_remove.comp:19-49 (active bricks; every cell is written: value or 0)
__global__ __launch_bounds__(BRICK_THREADS) void k_divergence_remove_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                       const int8_t* __restrict__ marker, const float* __restrict__ p, const float4* __restrict__ solid,
                                                                       float* __restrict__ vx, float* __restrict__ vy, float* __restrict__ vz) {
    const Grid g = bg.g;
    BRICK_LOOP_BEGIN(bg, list, count)
        QuadMarkers m; m.c = *reinterpret_cast<const uint32_t*>(marker + base);
        load_quad_markers(marker, g, base, x0, y, z, m);
        const bool any = any_fluid4(m.c) || m.xp == CELL_FLUID || any_fluid4(m.yp) || any_fluid4(m.zp);
        float ox[4] = {0.f, 0.f, 0.f, 0.f}, oy[4] = {0.f, 0.f, 0.f, 0.f}, oz[4] = {0.f, 0.f, 0.f, 0.f};
        if (any) {
            const int plane = g.nx * g.ny;
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 pc = ld4(p + base);
            const float pxp = x0 + 4 < g.nx ? p[base + 4] : 0.0f;
            const float4 pyp = y + 1 < g.ny ? ld4(p + base + g.nx) : zero, pzp = z + 1 < g.nz ? ld4(p + base + plane) : zero;
            const float4 vxc = ld4(vx + base), vyc = ld4(vy + base), vzc = ld4(vz + base);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + j;
                const int mc = mbyte(m.c, j);
                const float pcj = (mc == CELL_FLUID) ? f4(pc, j) : 0.0f;
                const int mn[3] = {j < 3 ? mbyte(m.c, j + 1) : m.xp, mbyte(m.yp, j), mbyte(m.zp, j)};
                const float pn[3] = {j < 3 ? f4(pc, j + 1) : pxp, f4(pyp, j), f4(pzp, j)};
                const float vc[3] = {f4(vxc, j), f4(vyc, j), f4(vzc, j)};
                float res[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = 0.0f;
                    if (mc == CELL_FLUID || mn[c] == CELL_FLUID) {
                        if (mc == CELL_SOLID) v = solid_comp(solid, g, x, y, z, c);
                        else if (mn[c] == CELL_SOLID) v = solid_comp(solid, g, x + (c == 0), y + (c == 1), z + (c == 2), c);
                        else { v = vc[c]; v -= pcj - ((mn[c] == CELL_FLUID) ? pn[c] : 0.0f); }
                    }
                    res[c] = v;
                }
                ox[j] = res[0]; oy[j] = res[1]; oz[j] = res[2];
            }
        }
        *reinterpret_cast<float4*>(vx + base) = make_float4(ox[0], ox[1], ox[2], ox[3]);
        *reinterpret_cast<float4*>(vy + base) = make_float4(oy[0], oy[1], oy[2], oy[3]);
        *reinterpret_cast<float4*>(vz + base) = make_float4(oz[0], oz[1], oz[2], oz[3]);
    BRICK_LOOP_END
}

// R2: density_projection_position_change.comp:18-51 (active bricks; every cell is written)
__global__ __launch_bounds__(BRICK_THREADS) void k_position_change_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                     const int8_t* __restrict__ marker, const float* __restrict__ p, float dt,
                                                                     float* __restrict__ vx, float* __restrict__ vy, float* __restrict__ vz) {
    const Grid g = bg.g;
    BRICK_LOOP_BEGIN(bg, list, count)
        QuadMarkers m; m.c = *reinterpret_cast<const uint32_t*>(marker + base);
        load_quad_markers(marker, g, base, x0, y, z, m);
        const bool any = any_fluid4(m.c) || m.xp == CELL_FLUID || any_fluid4(m.yp) || any_fluid4(m.zp);
        float ox[4] = {0.f, 0.f, 0.f, 0.f}, oy[4] = {0.f, 0.f, 0.f, 0.f}, oz[4] = {0.f, 0.f, 0.f, 0.f};
        if (any) {
            const int plane = g.nx * g.ny;
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 pc = ld4(p + base);
            const float pxp = x0 + 4 < g.nx ? p[base + 4] : 0.0f;
            const float4 pyp = y + 1 < g.ny ? ld4(p + base + g.nx) : zero, pzp = z + 1 < g.nz ? ld4(p + base + plane) : zero;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mc = mbyte(m.c, j);
                const float pcj = (mc == CELL_FLUID) ? f4(pc, j) : 0.0f;
                const int mn[3] = {j < 3 ? mbyte(m.c, j + 1) : m.xp, mbyte(m.yp, j), mbyte(m.zp, j)};
                const float pn[3] = {j < 3 ? f4(pc, j + 1) : pxp, f4(pyp, j), f4(pzp, j)};
                float res[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float d = (((mn[c] == CELL_FLUID) ? pn[c] : 0.0f) - pcj) * dt;
                    if (mc == CELL_SOLID || mn[c] == CELL_SOLID) d = 0.0f;
                    res[c] = d;
                }
                ox[j] = res[0]; oy[j] = res[1]; oz[j] = res[2];
            }
        }
        *reinterpret_cast<float4*>(vx + base) = make_float4(ox[0], ox[1], ox[2], ox[3]);
        *reinterpret_cast<float4*>(vy + base) = make_float4(oy[0], oy[1], oy[2], oy[3]);
        *reinterpret_cast<float4*>(vz + base) = make_float4(oz[0], oz[1], oz[2], oz[3]);
    BRICK_LOOP_END
}

// D3: extrapolate_velocity.comp:9-90 (active bricks).  Every marker test of the shader is "== FLUID", so the block stages its brick plus a
// one-cell ring (18 x 10 x 6 cells) as ONE BIT per cell: 60 threads each turn a 24-byte marker row (x0-4 .. x0+19) into a 24-bit FLUID mask.
// A quad's whole 12 x 3 x 3 neighbourhood is then nine shifted row masks in registers, "face has a FLUID side" is an OR of two masks and
// the shader's skip conditions are bit tests -- the byte-wise version spent ~1 900 VALU instructions per thread on marker extraction and
// was issue-bound.  Bricks whose tile holds no FLUID cell are skipped after the load (most of the dilated ring).
constexpr int ET_ROW = 24;                                   // bytes per staged row: x0-4 .. x0+19 (6 dwords)
constexpr int ET_ROWS = (BY + 2) * (BZ + 2);                 // 60 rows
__global__ __launch_bounds__(BRICK_THREADS) void k_extrapolate_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                 const int8_t* __restrict__ marker, float* __restrict__ vx, float* __restrict__ vy, float* __restrict__ vz) {
    __shared__ uint32_t rowmask[2][ET_ROWS + 4];             // double buffered by brick parity: one barrier per brick
    // the mean over n = 1..8 valid neighbour faces as a correctly rounded division by a CONSTANT: n = m 2^k, m in {1, 3, 5, 7}; scaling by 2^-k
    // is exact and q0 = RN(y c), q = fma(fma(-m, q0, y), c, q0) with c = RN(1/m) is RN(y / m) (checked for every f32 significand by
    // tests/native/div_const_check.c): bit-identical to `avg / num`, a third of the instructions
    __shared__ float4 divn[9];
    if (threadIdx.x < 9) {
        const int nn = (int)threadIdx.x;
        float c = 1.0f, nm = -1.0f, sc = 1.0f;
        if (nn == 3 || nn == 6) { c = 0x1.555556p-2f; nm = -3.0f; }
        if (nn == 5) { c = 0x1.99999ap-3f; nm = -5.0f; }
        if (nn == 7) { c = 0x1.24924ap-3f; nm = -7.0f; }
        if (nn == 2 || nn == 6) sc = 0.5f;
        if (nn == 4) sc = 0.25f;
        if (nn == 8) sc = 0.125f;
        divn[nn] = make_float4(c, nm, sc, 0.0f);
    }
    __syncthreads();
    const Grid g = bg.g;
    float* vel[3] = {vx, vy, vz};
    const uint32_t n = *count;
    const int plane = g.nx * g.ny;
    int parity = 0;
    for (uint32_t j = blockIdx.x; j < list_slots(n); j += gridDim.x) {
        uint32_t i;
        if (!list_slot(j, n, i)) continue;
        parity ^= 1;
        const uint32_t b = list[i] & ~STALE_BIT;
        int bx, by, bz; brick_coords(bg, b, bx, by, bz);
        const int tx0 = bx * BX - 4, ty0 = by * BY - 1, tz0 = bz * BZ - 1;     // tile origin (x is dword aligned)
        uint32_t* rm = rowmask[parity];
        bool any = false;
        if (threadIdx.x < ET_ROWS) {
            const int row = threadIdx.x;
            const int yy = ty0 + row % (BY + 2), zz = tz0 + row / (BY + 2);
            uint32_t v[ET_ROW / 4];
#pragma unroll
            for (int dw = 0; dw < ET_ROW / 4; ++dw) {
                const int xx = tx0 + dw * 4;
                v[dw] = 0;   // out of bounds reads SOLID (0)
                if ((unsigned)yy < (unsigned)g.ny && (unsigned)zz < (unsigned)g.nz && xx >= 0 && xx < g.nx) v[dw] = *reinterpret_cast<const uint32_t*>(marker + cidx(g, xx, yy, zz));
            }
            uint32_t mask = 0;
#pragma unroll
            for (int dw = 0; dw < ET_ROW / 4; ++dw) {
                // marker bytes are 0x00 SOLID, 0x01 FLUID, 0xFF AIR: FLUID <=> bit 0 set and bit 7 clear; the multiply gathers the four bits
                const uint32_t f = v[dw] & ~(v[dw] >> 7) & 0x01010101u;
                mask |= ((f * 0x10204080u) >> 28) << (4 * dw);
            }
            rm[row] = mask;
            any = mask != 0;
        }
        if (!__syncthreads_or(any)) continue;     // (the barrier also publishes the masks; the next brick writes the other buffer)
        int x0, y, z;
        const bool valid = brick_quad(bg, b, threadIdx.x, x0, y, z);
        if (!valid) continue;
        // F[dz][dy]: bit k = cell (x0 - 4 + k, y + dy - 1, z + dz - 1) is FLUID, k = 0 .. 11; the quad's cells are k = 4 .. 7
        const int sh = x0 - tx0 - 4, r0 = (z - 1 - tz0) * (BY + 2) + (y - 1 - ty0);
        uint32_t F[3][3];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) F[dz][dy] = (rm[r0 + dz * (BY + 2) + dy] >> sh) & 0xFFFu;
        uint32_t all = 0;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) all |= F[dz][dy];
        if (!(all & 0x1F8u)) continue;            // no FLUID cell in x0-1 .. x0+4 of the 3 x 3 rows: nothing to extrapolate from
        // face masks: bit k = the face of cell k towards +comp has a FLUID side (extrapolate_velocity.comp:30-37 validity test)
        uint32_t X[3][3], Y[3], Z[3];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) X[dz][dy] = F[dz][dy] | (F[dz][dy] >> 1);
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) Y[dz] = F[dz][1] | F[dz][2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) Z[dy] = F[1][dy] | F[2][dy];
        // cells that take part per component: not FLUID, own face without a FLUID side, at least one valid in-plane neighbour face
        uint32_t anyok[3];
        anyok[0] = X[0][0] | X[0][1] | X[0][2] | X[1][0] | X[1][2] | X[2][0] | X[2][1] | X[2][2];
        anyok[1] = (Y[0] >> 1) | Y[0] | (Y[0] << 1) | (Y[1] >> 1) | (Y[1] << 1) | (Y[2] >> 1) | Y[2] | (Y[2] << 1);
        anyok[2] = (Z[0] >> 1) | Z[0] | (Z[0] << 1) | (Z[1] >> 1) | (Z[1] << 1) | (Z[2] >> 1) | Z[2] | (Z[2] << 1);
        const uint32_t own[3] = {X[1][1], Y[1], Z[1]};
        const int base = cidx(g, x0, y, z);
        const bool ym_in = y > 0, yp_in = y + 1 < g.ny, zm_in = z > 0, zp_in = z + 1 < g.nz;
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) {
            const uint32_t need = ~F[1][1] & ~own[comp] & anyok[comp] & 0xF0u;
            if (!need) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 + j;
                if (!((need >> k) & 1u)) continue;
                const int x = x0 + j;
                // the 8 in-plane neighbour faces in the reference's order (b2 outer, a inner): validity bits from the masks; every face is
                // LOADED (an invalid or outside one from the cell itself: same cache lines, no exec-mask juggling per load) and masked to 0
                uint32_t okm = 0; float val[8]; int k8 = 0;
#pragma unroll
                for (int b2 = -1; b2 <= 1; ++b2)
#pragma unroll
                    for (int a = -1; a <= 1; ++a) {
                        if (a == 0 && b2 == 0) continue;
                        int ox, oy, oz; uint32_t fm;
                        if (comp == 0) { ox = 0; oy = a; oz = b2; fm = X[1 + b2][1 + a]; }
                        else if (comp == 1) { ox = a; oy = 0; oz = b2; fm = Y[1 + b2]; }
                        else { ox = a; oy = b2; oz = 0; fm = Z[1 + b2]; }
                        const uint32_t ok = (fm >> (k + ox)) & 1u;
                        okm |= ok << k8;
                        const bool in = (ox < 0 ? x > 0 : ox > 0 ? x + 1 < g.nx : true) && (oy < 0 ? ym_in : oy > 0 ? yp_in : true) && (oz < 0 ? zm_in : oz > 0 ? zp_in : true);
                        const bool use = ok != 0u && in;                        // (faces outside the grid read 0 but still count when their inner side is FLUID)
                        const uint32_t cc = (uint32_t)(use ? base + j + ox + oy * g.nx + oz * plane : base + j);
                        const float v = ld1o(vel[comp], cc * 4u);
                        val[k8] = use ? v : 0.0f;
                        ++k8;
                    }
                float avgV = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; ++q) avgV += val[q];                     // invalid faces add an exact 0 (the reference skips them)
                const float4 dc = divn[__popc(okm)];                            // >= 1: `need` guarantees a valid face
                const float y = avgV * dc.z, q0 = y * dc.x;
                st1o(vel[comp], (uint32_t)(base + j) * 4u, fmaf(fmaf(dc.y, q0, y), dc.x, q0));
            }
        }
    }
}

}  // namespace blubk
