// Device kernels of the blub fluid step for gfx950 (MI355X, wave64).  Included by blub_fluid.hip only.
//
// Every kernel cites the reference shader it re-implements (paths relative to /root/reference/shader/simulation).
// Arithmetic is written in the same operation order as the reference GLSL and compiled with -ffp-contract=off, so
// that element-wise kernels are bit-reproducible against the CPU oracle; only summation ORDER (linked-list order,
// dot-product trees) differs and is covered by the stated tolerances.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blubk {

constexpr int CELL_SOLID = 0, CELL_FLUID = 1, CELL_AIR = -1;
constexpr uint32_t INVALID_LL = 0xFFFFFFFFu;

struct Grid { int nx, ny, nz; };

// Particle kernels take their particle count from the host argument or -- z-slab groups, whose counts change on the device without the host
// looking (blub_slab.inc.hip) -- from device memory: n_dev = {own particles, ghost particles}, sel bit 0 / 1 = count the own / the ghosts;
// the host argument is then only the bound the launch grid was sized for.
__device__ __forceinline__ uint32_t particle_count(uint32_t n_host, const uint32_t* __restrict__ n_dev, uint32_t sel) {
    if (!n_dev) return n_host;
    const uint32_t n = ((sel & 1u) ? n_dev[0] : 0u) + ((sel & 2u) ? n_dev[1] : 0u);
    return n < n_host ? n : n_host;
}
__device__ __forceinline__ bool inb(const Grid& g, int x, int y, int z) {
    return (unsigned)x < (unsigned)g.nx && (unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz;
}
__device__ __forceinline__ int cidx(const Grid& g, int x, int y, int z) { return (z * g.ny + y) * g.nx + x; }
// texelFetch / imageLoad semantics: out of bounds reads 0
__device__ __forceinline__ int mk(const int8_t* __restrict__ m, const Grid& g, int x, int y, int z) { return inb(g, x, y, z) ? (int)m[cidx(g, x, y, z)] : CELL_SOLID; }
__device__ __forceinline__ float fv(const float* __restrict__ v, const Grid& g, int x, int y, int z) { return inb(g, x, y, z) ? v[cidx(g, x, y, z)] : 0.0f; }
__device__ __forceinline__ float satf(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float fractf(float v) { return v - floorf(v); }
__device__ __forceinline__ float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float comp3(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }

// ---- wave64 / block reductions (wavefront shuffles; deterministic order) ---------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    return v;
}
// Sum over a 256-thread block; result valid in every thread. `sm` needs 8 floats.
template <bool MAX>
__device__ __forceinline__ float block_reduce_256(float v, float* sm) {
    v = MAX ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = sm[0];
    if (MAX) { r = fmaxf(r, sm[1]); r = fmaxf(r, sm[2]); r = fmaxf(r, sm[3]); }
    else { r += sm[1]; r += sm[2]; r += sm[3]; }
    return r;
}
// Every block re-reduces the per-block partials of the previous kernel (n <= a few thousand floats, L2 resident):
// no atomics, no extra launch, bit-deterministic.
template <bool MAX>
__device__ __forceinline__ float reduce_partials_256(const float* __restrict__ part, int n, float* sm) {
    float v = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) v = MAX ? fmaxf(v, part[i]) : v + part[i];
    return block_reduce_256<MAX>(v, sm);
}

// =================================================================================================================
// T2 x3 fused: transfer_build_linkedlist.comp:10-26.  One pass over the particles builds the three staggered
// dual-grid lists (component c: dual cell = ivec3(pos - (0.5 + 0.5 e_c))) and marks FLUID cells.
// list "next" pointers: component x lives in pos.w (as in the reference), y/z in two extra u32 arrays.
// =================================================================================================================
// Wave-aggregated list insertion.  A per-lane atomicExch costs a device atomic per particle and list (2.9 M per step on the headline scene:
// measured 58 us with the particles freshly binned, 114 us sixty steps later, against 36 us for 1/8 of them).  Instead ALL lanes of the wave
// that insert into the same list are chained in registers, in lane order (lane -> the previous lane of its group), and only the group's LAST
// lane exchanges the list head; the old head becomes the `next` of the group's FIRST lane.  The result is a valid insertion order of the
// reference's atomic-exchange list (transfer_build_linkedlist.comp:25).  Groups are found with one ballot per distinct key (a scalar loop,
// 8-16 rounds for 64 particles of neighbouring cells): unlike runs of ADJACENT equal lanes they survive the decay of the particle order
// between two rebinnings.  key < 0: the particle is outside the grid (no insertion, next = invalid).  All 64 lanes must call this.
__device__ __forceinline__ uint32_t wave_list_insert(uint32_t* __restrict__ heads, int key, uint32_t particle) {
    const int lane = threadIdx.x & 63;
    // (capping the number of rounds and letting the left-over lanes insert on their own was measured: 16 rounds 77 us, 32 rounds 67 us, no cap 62-65 us
    // per step -- the device atomics are what costs, not the search)
    unsigned long long remaining = ~0ull, mine = 0ull;
    while (remaining) {
        const int k = __builtin_amdgcn_readlane(key, __builtin_ctzll(remaining));     // uniform source lane: v_readlane (__shfl would be a ds_bpermute + wait per round)
        const unsigned long long m = __ballot(key == k);
        if (key == k) mine = m;
        remaining &= ~m;
    }
    const unsigned long long below = mine & ((1ull << lane) - 1ull);
    const bool is_first = below == 0ull, is_last = (mine >> lane) == 1ull;
    const int prev_lane = is_first ? lane : 63 - __builtin_clzll(below), last_lane = 63 - __builtin_clzll(mine);
    uint32_t old = 0;
    if (is_last && key >= 0) old = atomicExch(heads + key, particle + 1);
    const uint32_t old_of_group = __shfl(old, last_lane, 64);
    const uint32_t prev_particle = __shfl(particle, prev_lane, 64);
    if (key < 0) return INVALID_LL;
    return is_first ? old_of_group - 1u : prev_particle;
}

// Wave-aggregated COUNTING insertion (the particle re-sort, blub_bricks.hip.h): the lanes of a wave that share a key take consecutive ranks in lane
// order behind ONE atomic add -- the group search of wave_list_insert.  In three parts: the group's lanes (wave_group), the add by its first lane
// (wave_group_add: the counter's old value, valid in that lane), the rank (wave_group_rank).  key < 0: no insertion.  All 64 lanes must call these.
__device__ __forceinline__ unsigned long long wave_group(int key) {
    unsigned long long remaining = ~0ull, mine = 0ull;
    while (remaining) {
        const int k = __builtin_amdgcn_readlane(key, __builtin_ctzll(remaining));
        const unsigned long long m = __ballot(key == k);
        if (key == k) mine = m;
        remaining &= ~m;
    }
    return mine;
}
__device__ __forceinline__ uint32_t wave_group_add(uint32_t* __restrict__ counts, int key, unsigned long long mine) {
    const int lane = threadIdx.x & 63;
    return ((mine & ((1ull << lane) - 1ull)) == 0ull && key >= 0) ? atomicAdd(counts + key, (uint32_t)__popcll(mine)) : 0u;
}
__device__ __forceinline__ uint32_t wave_group_rank(uint32_t base, unsigned long long mine) {
    const int lane = threadIdx.x & 63;
    return (uint32_t)__shfl((int)base, __builtin_ctzll(mine), 64) + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
}

// Gather nodes: what one hop of a P2G list walk reads, in ONE 32-byte piece -- {position, link of component c's list, velocity row c}.  One array of
// nodes per component, indexed by particle: a node never straddles a 64-byte sector, consecutive particles share sectors and lines.
// (The walk used to read position, row and link from three arrays: three sectors per hop, 36 useful bytes of 192.)
struct alignas(16) GatherNode { float px, py, pz; uint32_t next; float4 row; };
static_assert(sizeof(GatherNode) == 32, "GatherNode");

__global__ __launch_bounds__(256) void k_build_lists(Grid g, uint32_t num_particles, float4* __restrict__ pos, int8_t* __restrict__ marker,
                                                     uint32_t* __restrict__ ll0, uint32_t* __restrict__ ll1, uint32_t* __restrict__ ll2,
                                                     const float4* __restrict__ pvx, const float4* __restrict__ pvy, const float4* __restrict__ pvz,
                                                     GatherNode* __restrict__ nodes, uint32_t node_stride, int no_solid_voxels, int write_links,
                                                     const uint32_t* __restrict__ n_dev, uint32_t n_sel) {
    num_particles = particle_count(num_particles, n_dev, n_sel);
    if (blockIdx.x * 256u >= num_particles) return;      // (uniform)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < num_particles;          // no early return: the wave-level insertion needs every lane
    float4 p = make_float4(-8.f, -8.f, -8.f, 0.f);
    float4 rows[3] = {p, p, p};
    if (live) {
        p = pos[i];
        rows[0] = pvx[i]; rows[1] = pvy[i]; rows[2] = pvz[i];
        const int x = (int)p.x, y = (int)p.y, z = (int)p.z;
        if (inb(g, x, y, z)) {
            const int c = cidx(g, x, y, z);
            // without solid voxels only the domain shell is SOLID: an interior cell is marked without reading the marker first
            const bool interior = no_solid_voxels && x > 0 && y > 0 && z > 0 && x < g.nx - 1 && y < g.ny - 1 && z < g.nz - 1;
            if (interior || marker[c] != CELL_SOLID) marker[c] = CELL_FLUID;
        }
    }
    uint32_t nxt[3];
    uint32_t* heads[3] = {ll0, ll1, ll2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int dx = (int)(p.x - (c == 0 ? 1.0f : 0.5f)), dy = (int)(p.y - (c == 1 ? 1.0f : 0.5f)), dz = (int)(p.z - (c == 2 ? 1.0f : 0.5f));
        const int key = (live && inb(g, dx, dy, dz)) ? cidx(g, dx, dy, dz) : -1;
        nxt[c] = wave_list_insert(heads[c], key, i);
    }
    // particles_position_ll keeps the x list's links, as in the reference (the whole record: a 4-byte store is a partial sector write).  Inside a step
    // nobody reads them before k_advect overwrites the record (the walks read the nodes): the stage hook asks for them, blub_fluid_step does not.
    if (live && write_links) pos[i] = make_float4(p.x, p.y, p.z, __uint_as_float(nxt[0]));
    // Nodes are stored per component (nodes[c * node_stride + i]): the nodes of consecutive particles -- after a rebinning the members of a list -- share
    // sectors and lines.  The wave's 64 x 32 bytes per component are one contiguous 2 KiB run: transposed through LDS so that every store instruction
    // writes 1 KiB of consecutive bytes (a lane storing its own float4s makes each instruction touch 64 different sectors: six times the write
    // requests at the L2).
    __shared__ float4 stage[4][64 * 6];
    float4* const st = stage[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        st[c * 128 + lane * 2] = make_float4(p.x, p.y, p.z, __uint_as_float(nxt[c]));
        st[c * 128 + lane * 2 + 1] = rows[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t wave_first = i - (uint32_t)lane;                                                  // (may lie beyond the particles for the last block's trailing waves)
    const uint32_t wave_quads = wave_first < num_particles ? min(64u, num_particles - wave_first) * 2u : 0u;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int c = k >> 1;
        const uint32_t e = (uint32_t)((k & 1) * 64 + lane);
        if (e < wave_quads) reinterpret_cast<float4*>(nodes + (size_t)c * node_stride)[2 * (size_t)wave_first + e] = st[c * 128 + e];
    }
}

// ---- shared by the P2G gather (blub_bricks.hip.h): transfer_gather_velocity.comp:18-26 ----
__device__ __forceinline__ void add_particle(float& v, float& wsum, const float4& pp, const float4& row, float sx, float sy, float sz) {
    const float tx = sx - pp.x, ty = sy - pp.y, tz = sz - pp.z;                       // :20
    const float ox = satf(1.0f - fabsf(tx)), oy = satf(1.0f - fabsf(ty)), oz = satf(1.0f - fabsf(tz));
    const float w = ox * oy * oz;                                                      // :22
    const float d = ((row.x * tx + row.y * ty) + row.z * tz) + row.w * 1.0f;           // :24
    v += w * d;
    wsum += w;
}

// =================================================================================================================
// PCG (pressure_solver.rs:591-729, shader/simulation/pressure_solver/*; schedule in SURVEY Appendix D).
//
// MI355X formulation: three streaming kernels per iteration instead of the reference's ~9 dispatches
//   K1  apply   : partial dot  s.As                                   (pressure_apply_coeff.comp + reduce level 1)
//   K2  update  : alpha from partials; p += alpha s; r -= alpha A s; [max|r|]; z = M^-1 r; partial z.r
//                 (pressure_update_pressure_and_residual.comp + apply_preconditioner x2 for the "zero" reading)
//   K5  search  : beta from partials; s = z + beta s; convergence bookkeeping (pressure_update_search.comp +
//                 pressure_reduce.comp's MAX_ERROR mode)
// The dot products never touch a 4N-byte reduce buffer: each block writes ONE partial, the consumer kernel's blocks
// re-reduce the <=PCG_GRID partials.  sigma is the partial array of the previous z.r (double buffered by iteration
// parity), so no scalar needs a dedicated pass.  `done` replaces the zeroed indirect-dispatch arguments.
//
// Work decomposition: a tile = 256 threads x 4 x-consecutive cells (one float4 each) of a z-plane, marched over
// PCG_ZC planes; a fixed grid of persistent blocks strides over the tiles; tiles without FLUID cells are skipped.
// =================================================================================================================

struct PcgCtrl {       // device-resident; [0..1] mirror the reference's MaxError / NumIterations read-back (pressure_init.comp:8-15)
    float max_err;
    float num_iter;
    int done;
    int pad;
    float sigma[2];    // sigma of iteration i in slot i & 1 (the reference keeps it in PcgScalars.Sigma, pressure.glsl:24-28)
    uint32_t seq;      // host-assigned solve number, written by the last kernel of the solve: tags the asynchronous read-back
    uint32_t pad2;
};

struct PcgGeom {
    Grid g;
    int qpr;          // quads per row  = nx/4
    int qpp;          // quads per plane
    int plane_blocks; // ceil(qpp/256)
    int z_chunks;
    int tiles;
    int zc;           // planes marched per tile
};

struct QuadMarkers { uint32_t c, ym, yp, zm, zp; int xm, xp; };

__device__ __forceinline__ int mbyte(uint32_t packed, int j) { return (int)(int8_t)((packed >> (8 * j)) & 0xFFu); }
__device__ __forceinline__ bool any_fluid4(uint32_t packed) {
    return mbyte(packed, 0) == CELL_FLUID || mbyte(packed, 1) == CELL_FLUID || mbyte(packed, 2) == CELL_FLUID || mbyte(packed, 3) == CELL_FLUID;
}
__device__ __forceinline__ void load_quad_markers(const int8_t* __restrict__ M, const Grid& g, int base, int x0, int y, int z, QuadMarkers& q) {
    const int plane = g.nx * g.ny;
    q.xm = x0 > 0 ? (int)M[base - 1] : CELL_SOLID;
    q.xp = x0 + 4 < g.nx ? (int)M[base + 4] : CELL_SOLID;
    q.ym = y > 0 ? *reinterpret_cast<const uint32_t*>(M + base - g.nx) : 0u;
    q.yp = y + 1 < g.ny ? *reinterpret_cast<const uint32_t*>(M + base + g.nx) : 0u;
    q.zm = z > 0 ? *reinterpret_cast<const uint32_t*>(M + base - plane) : 0u;
    q.zp = z + 1 < g.nz ? *reinterpret_cast<const uint32_t*>(M + base + plane) : 0u;
}
struct QuadValues { float4 c, ym, yp, zm, zp; float xm, xp; };
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// 32-bit byte offsets from a uniform base pointer: the compiler addresses these as `global_load v, v_off, s[base]` (no 64-bit VALU
// address arithmetic per access; an f32 volume is < 4 GiB for every supported grid)
__device__ __forceinline__ float4 ld4o(const float* base, uint32_t byte_off) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off); }
__device__ __forceinline__ float ld1o(const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); }
__device__ __forceinline__ void st4o(float* base, uint32_t byte_off, const float4& v) { *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v; }

// ---- data another AGENT reads or writes while a kernel runs (z-slab groups, direct transport: blub_slab.hip.h) -----------------------
// Write-through stores and cache-bypassing loads (sc0 sc1: system scope) instead of release / acquire fences, which cost a write-back +
// invalidate of the whole L2 per workgroup (profiles/r03_tree_barrier_probe.txt: the recipe of cdna_hip_programming.md G16).  A producer
// makes its write-through stores, waits for them (s_waitcnt vmcnt(0)), and only then raises a flag with st_sys_u32; a consumer polls the flag
// with ld_sys_u32 and reads the payload with ld_sys_*.  Nothing else about these addresses may be cached by the consumer: it never reads
// them with plain loads.
typedef float blub_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sys_f4(float4* p, const float4& a) {
    blub_v4f v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w;
    // (s_nop: a store of more than 64 bits followed by a VALU write of its data registers needs a wait state the compiler inserts for its own
    //  stores but not behind inline assembly -- without it lane 1 of the float4 came out overwritten by the next instruction's result)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sys_u32(uint32_t* p, uint32_t a) { asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(a) : "memory"); }
__device__ __forceinline__ void st_sys_u8(uint8_t* p, uint32_t a) { asm volatile("global_store_byte %0, %1, off sc0 sc1" :: "v"(p), "v"(a) : "memory"); }
__device__ __forceinline__ float4 ld_sys_f4(const float4* p) {
    blub_v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t ld_sys_u32(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// several cache-bypassing loads in flight, ONE wait: issue with ld_sys_f4_issue, then ld_sys_wait on the same registers before the first use
__device__ __forceinline__ void ld_sys_f4_issue(blub_v4f& v, const float4* p) { asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void ld_sys_wait(blub_v4f& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a) :: "memory"); }
__device__ __forceinline__ void ld_sys_wait(blub_v4f& a, blub_v4f& b, blub_v4f& c, blub_v4f& d) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory"); }

// What a kernel of one slab needs to exchange with the other slabs by itself (direct transport of a z-slab group): pointers into the
// PEERS' memory (the same process in a loopback group, hipIpc mappings between processes) for what it produces, its own flag words for what
// it consumes.  Flags carry the sequence number of the exchange (the host numbers them identically on every rank; >= compares).
constexpr int SLAB_MAX_PEERS = 7;
struct SlabDirect {
    float* w_up; float* w_dn;                 // the z-neighbours' copies of the field whose boundary planes this launch produces (global grid coordinates); nullptr: none
    float* p_up; float* p_dn;                 // ... and of the pressure field
    float4* part_out[SLAB_MAX_PEERS];         // my segment of every other slab's partial array (this launch's parity)
    uint32_t* flag_out[SLAB_MAX_PEERS];       // every other slab's flag word for messages from me
    const uint32_t* flags_in;                 // my flag words, one per source rank
    uint32_t* blocks_done;                    // my counter of finished workgroups (last one raises the flags and resets it)
    uint32_t* error;                          // set when a wait timed out (bounded spins: a missing peer must not hang the GPU)
    uint32_t seq_in, seq_out, wait_mask;      // wait for flags_in[r] >= seq_in for every bit r of wait_mask; publish seq_out
    int n_out;
    float4* log;                              // diagnostic (any transport, also the single domain; nullptr = off): entry i = {gamma_i, delta_i, max|r_i|, alpha_i} as K(i) reduced them
    unsigned long long* stamps;               // diagnostic (nullptr = off; "pcg_phase_stamps" tuning): K(i)'s workgroup 0 leaves 8 time stamps (s_memrealtime, 10 ns ticks) at its phase boundaries in entry i
};
// one time stamp of the intra-kernel timeline (blub_fluid_read_phase_stamps): workgroup 0, thread 0 only
__device__ __forceinline__ void phase_stamp(const SlabDirect* dir, int iteration, int k) {
    if (dir && dir->stamps && blockIdx.x == 0 && threadIdx.x == 0 && iteration >= 0 && iteration < 64) dir->stamps[iteration * 8 + k] = __builtin_amdgcn_s_memrealtime();
}
constexpr unsigned SLAB_SPIN_LIMIT = 1u << 24;
__device__ __forceinline__ bool slab_gave_up(const uint32_t* error) { return error && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
// every thread of the block returns once all awaited flags have arrived (or the bounded wait ran out)
__device__ __forceinline__ void slab_wait_flags(const uint32_t* flags_in, uint32_t mask, uint32_t seq, uint32_t* error) {
    if (threadIdx.x < 32 && ((mask >> threadIdx.x) & 1u)) {
        unsigned spins = 0;
        while ((int32_t)(ld_sys_u32(flags_in + threadIdx.x) - seq) < 0) {
            __builtin_amdgcn_s_sleep(2);
            // (~8 s: ranks may be seconds apart at start-up; once ANY wait of this slab has run out -- a peer stopped stepping -- the later ones give up at once)
            if (++spins > SLAB_SPIN_LIMIT || ((spins & 1023u) == 0u && slab_gave_up(error))) { if (error) atomicOr(error, 1u); break; }
        }
    }
    __syncthreads();
}
// after a block's write-through stores: the LAST block to finish raises the flags
__device__ __forceinline__ void slab_publish(const SlabDirect& D, uint32_t participating_blocks) {
    wait_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(D.blocks_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1u == participating_blocks) {
            __hip_atomic_store(D.blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int q = 0; q < D.n_out; ++q) st_sys_u32(D.flag_out[q], D.seq_out);
        }
    }
}

__device__ __forceinline__ void st1o(float* base, uint32_t byte_off, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v; }
__device__ __forceinline__ uint32_t ldu32o(const uint8_t* base, uint32_t byte_off) { return *reinterpret_cast<const uint32_t*>(base + byte_off); }
__device__ __forceinline__ void load_quad_values(const float* __restrict__ S, const Grid& g, int base, int x0, int y, int z, QuadValues& v) {
    const int plane = g.nx * g.ny;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    v.c = ld4(S + base);
    v.xm = x0 > 0 ? S[base - 1] : 0.0f;
    v.xp = x0 + 4 < g.nx ? S[base + 4] : 0.0f;
    v.ym = y > 0 ? ld4(S + base - g.nx) : zero;
    v.yp = y + 1 < g.ny ? ld4(S + base + g.nx) : zero;
    v.zm = z > 0 ? ld4(S + base - plane) : zero;
    v.zp = z + 1 < g.nz ? ld4(S + base + plane) : zero;
}
__device__ __forceinline__ float f4(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// MultiplyWithCoefficientMatrix (pressure.glsl:34-75) for cell j of a quad; also returns d = #non-solid neighbours.
__device__ __forceinline__ float quad_mulA(const QuadMarkers& m, const QuadValues& v, int j, float& d) {
    const int mX0 = j > 0 ? mbyte(m.c, j - 1) : m.xm, mX1 = j < 3 ? mbyte(m.c, j + 1) : m.xp;
    const int mY0 = mbyte(m.ym, j), mY1 = mbyte(m.yp, j), mZ0 = mbyte(m.zm, j), mZ1 = mbyte(m.zp, j);
    d = (float)(mX0 != 0) + (float)(mX1 != 0) + (float)(mY0 != 0) + (float)(mY1 != 0) + (float)(mZ0 != 0) + (float)(mZ1 != 0);
    float r = 0.0f;
    r += d * f4(v.c, j);
    if (mX0 == CELL_FLUID) r -= (j > 0 ? f4(v.c, j - 1) : v.xm);
    if (mX1 == CELL_FLUID) r -= (j < 3 ? f4(v.c, j + 1) : v.xp);
    if (mY0 == CELL_FLUID) r -= f4(v.ym, j);
    if (mY1 == CELL_FLUID) r -= f4(v.yp, j);
    if (mZ0 == CELL_FLUID) r -= f4(v.zm, j);
    if (mZ1 == CELL_FLUID) r -= f4(v.zp, j);
    return r;
}
__device__ __forceinline__ float eps_div(float num, float den) { return num / (den + (den < 0.0f ? -1e-10f : 1e-10f)); }   // pressure_reduce.comp:71-77

#define PCG_TILE_LOOP_BEGIN(geom)                                                                            \
    for (int tile = blockIdx.x; tile < (geom).tiles; tile += gridDim.x) {                                    \
        const int pb = tile % (geom).plane_blocks, zc = tile / (geom).plane_blocks;                          \
        const int q = pb * 256 + threadIdx.x;                                                                \
        const bool qvalid = q < (geom).qpp;                                                                  \
        const int x0 = (q % (geom).qpr) << 2, y = q / (geom).qpr;                                            \
        const int z_begin = zc * (geom).zc, z_end = min(z_begin + (geom).zc, (geom).g.nz);
#define PCG_TILE_LOOP_END }

// The kernels below are the LITERAL kernel sequence of the LOD0 preconditioner reading (dense rows, marker based; stage_solve_lod0).
// S0 (pressure_init.comp:19-84): only r and p are updated (the generic preconditioner passes follow).
__global__ __launch_bounds__(256) void k_pcg_init(PcgGeom geom, const int8_t* __restrict__ marker, float* __restrict__ p, float* __restrict__ r,
                                                  float* __restrict__ s, float* __restrict__ part_sigma, uint8_t* __restrict__ tile_flags) {
    __shared__ float sm[8];
    __shared__ int s_any;
    float acc = 0.0f;
    PCG_TILE_LOOP_BEGIN(geom)
        if (threadIdx.x == 0) s_any = 0;
        __syncthreads();
        bool any = false;
        if (qvalid) {
            for (int z = z_begin; z < z_end; ++z) {
                const int base = cidx(geom.g, x0, y, z);
                const uint32_t mc = *reinterpret_cast<const uint32_t*>(marker + base);
                float4 pc = ld4(p + base);
                if (any_fluid4(mc)) {
                    any = true;
                    QuadMarkers m; m.c = mc; load_quad_markers(marker, geom.g, base, x0, y, z, m);
                    QuadValues pv; load_quad_values(p, geom.g, base, x0, y, z, pv);
                    float4 rc = ld4(r + base);
                    float rr[4] = {rc.x, rc.y, rc.z, rc.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (mbyte(mc, j) != CELL_FLUID) continue;
                        const int mX0 = j > 0 ? mbyte(m.c, j - 1) : m.xm, mX1 = j < 3 ? mbyte(m.c, j + 1) : m.xp;
                        const int mY0 = mbyte(m.ym, j), mY1 = mbyte(m.yp, j), mZ0 = mbyte(m.zm, j), mZ1 = mbyte(m.zp, j);
                        const float d = (float)(mX0 != 0) + (float)(mX1 != 0) + (float)(mY0 != 0) + (float)(mY1 != 0) + (float)(mZ0 != 0) + (float)(mZ1 != 0);
                        float res = rr[j];
                        if (d > 0.0f) res -= d * f4(pv.c, j);                                   // :62-63
                        if (mX0 == CELL_FLUID) res += (j > 0 ? f4(pv.c, j - 1) : pv.xm);          // :64-81
                        if (mX1 == CELL_FLUID) res += (j < 3 ? f4(pv.c, j + 1) : pv.xp);
                        if (mY0 == CELL_FLUID) res += f4(pv.ym, j);
                        if (mY1 == CELL_FLUID) res += f4(pv.yp, j);
                        if (mZ0 == CELL_FLUID) res += f4(pv.zm, j);
                        if (mZ1 == CELL_FLUID) res += f4(pv.zp, j);
                        rr[j] = res;
                    }
                    *reinterpret_cast<float4*>(r + base) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                }
                // pressure_init.comp:45-48: p := 0 outside the fluid
                bool dirty = false;
                if (mbyte(mc, 0) != CELL_FLUID && pc.x != 0.0f) { pc.x = 0.0f; dirty = true; }
                if (mbyte(mc, 1) != CELL_FLUID && pc.y != 0.0f) { pc.y = 0.0f; dirty = true; }
                if (mbyte(mc, 2) != CELL_FLUID && pc.z != 0.0f) { pc.z = 0.0f; dirty = true; }
                if (mbyte(mc, 3) != CELL_FLUID && pc.w != 0.0f) { pc.w = 0.0f; dirty = true; }
                if (dirty) *reinterpret_cast<float4*>(p + base) = pc;
            }
        }
        if (any) s_any = 1;
        __syncthreads();
        if (threadIdx.x == 0) tile_flags[tile] = (uint8_t)s_any;
        __syncthreads();
    PCG_TILE_LOOP_END
    const float tot = block_reduce_256<false>(acc, sm);
    if (threadIdx.x == 0 && part_sigma) part_sigma[blockIdx.x] = tot;
}

// K1: pressure_apply_coeff.comp:19-30 -- partial of s.As
__global__ __launch_bounds__(256) void k_pcg_apply(PcgGeom geom, const int8_t* __restrict__ marker, const float* __restrict__ s,
                                                   float* __restrict__ part_sas, const uint8_t* __restrict__ tile_flags, const PcgCtrl* __restrict__ ctrl) {
    __shared__ float sm[8];
    if (ctrl->done) return;
    float acc = 0.0f;
    PCG_TILE_LOOP_BEGIN(geom)
        if (!tile_flags[tile] || !qvalid) continue;
        for (int z = z_begin; z < z_end; ++z) {
            const int base = cidx(geom.g, x0, y, z);
            QuadMarkers m; m.c = *reinterpret_cast<const uint32_t*>(marker + base);
            if (!any_fluid4(m.c)) continue;
            load_quad_markers(marker, geom.g, base, x0, y, z, m);
            QuadValues sv; load_quad_values(s, geom.g, base, x0, y, z, sv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mbyte(m.c, j) != CELL_FLUID) continue;
                float d; const float as = quad_mulA(m, sv, j, d);
                acc += f4(sv.c, j) * as;
            }
        }
    PCG_TILE_LOOP_END
    const float tot = block_reduce_256<false>(acc, sm);
    if (threadIdx.x == 0) part_sas[blockIdx.x] = tot;
}

// K2: pressure_update_pressure_and_residual.comp:23-59
__global__ __launch_bounds__(256) void k_pcg_update(PcgGeom geom, const int8_t* __restrict__ marker, const float* __restrict__ s, float* __restrict__ p,
                                                    float* __restrict__ r, const float* __restrict__ part_sas, const float* __restrict__ part_sigma,
                                                    float* __restrict__ part_sigma_next, float* __restrict__ part_max, int num_part,
                                                    const uint8_t* __restrict__ tile_flags, const PcgCtrl* __restrict__ ctrl) {
    __shared__ float sm[8];
    if (ctrl->done) return;
    const float sigma = reduce_partials_256<false>(part_sigma, num_part, sm);
    const float sas = reduce_partials_256<false>(part_sas, num_part, sm);
    const float alpha = eps_div(sigma, sas);                                           // RESULTMODE_ALPHA
    float acc = 0.0f, emax = 0.0f;
    PCG_TILE_LOOP_BEGIN(geom)
        if (!tile_flags[tile] || !qvalid) continue;
        for (int z = z_begin; z < z_end; ++z) {
            const int base = cidx(geom.g, x0, y, z);
            QuadMarkers m; m.c = *reinterpret_cast<const uint32_t*>(marker + base);
            if (!any_fluid4(m.c)) continue;
            load_quad_markers(marker, geom.g, base, x0, y, z, m);
            QuadValues sv; load_quad_values(s, geom.g, base, x0, y, z, sv);
            float4 pc = ld4(p + base), rc = ld4(r + base);
            float pp[4] = {pc.x, pc.y, pc.z, pc.w}, rr[4] = {rc.x, rc.y, rc.z, rc.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mbyte(m.c, j) != CELL_FLUID) continue;
                float d; const float as = quad_mulA(m, sv, j, d);
                pp[j] = pp[j] + alpha * f4(sv.c, j);                                   // :39-40
                float res = rr[j];
                res -= alpha * as;                                                      // :52
                rr[j] = res;
                emax = fmaxf(emax, fabsf(res));                                         // :55
            }
            *reinterpret_cast<float4*>(p + base) = make_float4(pp[0], pp[1], pp[2], pp[3]);
            *reinterpret_cast<float4*>(r + base) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        }
    PCG_TILE_LOOP_END
    const float tot = block_reduce_256<false>(acc, sm);
    const float mx = block_reduce_256<true>(emax, sm);
    (void)tot; (void)part_sigma_next;
    if (threadIdx.x == 0) part_max[blockIdx.x] = mx;
}

// Generic preconditioner pass for the LOD0 reading: pressure_apply_preconditioner.comp:36-82 (both passes use the
// lower neighbours, Q3).  dot != nullptr => also emits the partial of out.r
__global__ __launch_bounds__(256) void k_pcg_precond_lod0(PcgGeom geom, const int8_t* __restrict__ marker, const float* __restrict__ in,
                                                          float* __restrict__ out, const float* __restrict__ rdot, float* __restrict__ part,
                                                          const uint8_t* __restrict__ tile_flags, const PcgCtrl* __restrict__ ctrl) {
    __shared__ float sm[8];
    if (ctrl->done) return;
    float acc = 0.0f;
    PCG_TILE_LOOP_BEGIN(geom)
        if (!tile_flags[tile] || !qvalid) continue;
        for (int z = z_begin; z < z_end; ++z) {
            const int base = cidx(geom.g, x0, y, z);
            QuadMarkers m; m.c = *reinterpret_cast<const uint32_t*>(marker + base);
            if (!any_fluid4(m.c)) continue;
            load_quad_markers(marker, geom.g, base, x0, y, z, m);
            QuadValues iv; load_quad_values(in, geom.g, base, x0, y, z, iv);
            float4 oc = ld4(out + base);
            float oo[4] = {oc.x, oc.y, oc.z, oc.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mbyte(m.c, j) != CELL_FLUID) continue;
                const int mX0 = j > 0 ? mbyte(m.c, j - 1) : m.xm, mX1 = j < 3 ? mbyte(m.c, j + 1) : m.xp;
                const int mY0 = mbyte(m.ym, j), mY1 = mbyte(m.yp, j), mZ0 = mbyte(m.zm, j), mZ1 = mbyte(m.zp, j);
                float res = f4(iv.c, j);
                if (mX0 == CELL_FLUID) res -= (j > 0 ? f4(iv.c, j - 1) : iv.xm);
                if (mY0 == CELL_FLUID) res -= f4(iv.ym, j);
                if (mZ0 == CELL_FLUID) res -= f4(iv.zm, j);
                const float d = (float)(mX0 != 0) + (float)(mX1 != 0) + (float)(mY0 != 0) + (float)(mY1 != 0) + (float)(mZ0 != 0) + (float)(mZ1 != 0);
                if (d > 0.0f) res /= d;
                oo[j] = res;
                if (rdot) acc += res * rdot[base + j];
            }
            *reinterpret_cast<float4*>(out + base) = make_float4(oo[0], oo[1], oo[2], oo[3]);
        }
    PCG_TILE_LOOP_END
    const float tot = block_reduce_256<false>(acc, sm);
    if (threadIdx.x == 0 && part) part[blockIdx.x] = tot;
}

// K5: pressure_update_search.comp:13-24 + the MAX_ERROR / BETA modes of pressure_reduce.comp:63-95.
// check: this iteration compared max|r| against the tolerance; last: i == max_num_iterations.
__global__ __launch_bounds__(256) void k_pcg_search(PcgGeom geom, const int8_t* __restrict__ marker, const float* __restrict__ r_or_z, float* __restrict__ s,
                                                    const float* __restrict__ part_sigma, const float* __restrict__ part_sigma_next,
                                                    const float* __restrict__ part_max, int num_part, const uint8_t* __restrict__ tile_flags,
                                                    PcgCtrl* __restrict__ ctrl, float tolerance, int iteration, int check, int last) {
    __shared__ float sm[8];
    if (ctrl->done) return;
    if (check) {
        const float err = reduce_partials_256<true>(part_max, num_part, sm);
        if (last || err < tolerance) {                                                  // pressure_reduce.comp:82-94
            if (blockIdx.x == 0 && threadIdx.x == 0) { ctrl->max_err = err; ctrl->num_iter = (float)iteration; ctrl->done = 1; }
            return;
        }
    }
    const float sigma = reduce_partials_256<false>(part_sigma, num_part, sm);
    const float sigma_next = reduce_partials_256<false>(part_sigma_next, num_part, sm);
    const float beta = eps_div(sigma_next, sigma);                                      // RESULTMODE_BETA
    PCG_TILE_LOOP_BEGIN(geom)
        if (!tile_flags[tile] || !qvalid) continue;
        for (int z = z_begin; z < z_end; ++z) {
            const int base = cidx(geom.g, x0, y, z);
            QuadMarkers m; m.c = *reinterpret_cast<const uint32_t*>(marker + base);
            if (!any_fluid4(m.c)) continue;
            const float4 zc4 = ld4(r_or_z + base);
            float4 sc = ld4(s + base);
            float ss[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mbyte(m.c, j) != CELL_FLUID) continue;
                const float zval = f4(zc4, j);
                ss[j] = zval + beta * ss[j];                                            // :23
            }
            *reinterpret_cast<float4*>(s + base) = make_float4(ss[0], ss[1], ss[2], ss[3]);
        }
    PCG_TILE_LOOP_END
}

// ---- samplers (SamplerPointClamp / SamplerTrilinearClamp with exact f32 weights) -------------
__device__ __forceinline__ float4 solid_point_clamp(const float4* __restrict__ solid, const Grid& g, float tx, float ty, float tz) {
    const int x = min(max((int)floorf(tx * (float)g.nx), 0), g.nx - 1);
    const int y = min(max((int)floorf(ty * (float)g.ny), 0), g.ny - 1);
    const int z = min(max((int)floorf(tz * (float)g.nz), 0), g.nz - 1);
    return solid[cidx(g, x, y, z)];
}
// What Vulkan leaves to the implementation is the ARITHMETIC of the linear filter (spec 16.8.3 gives the weighted sum, hardware evaluates it with
// fixed-point weights).  FILTER selects it (blub_fluid_set_filter_mode; the three evaluations of the shim that runs the reference's shaders, oracle/glsl/):
//   0 separable : lerps along x, then y, then z in f32 (what the oracle does; the default)
//   1 weighted  : the spec's formula in f32, tau = (1-a)(1-b)(1-g) t000 + a(1-b)(1-g) t100 + ..., summed in that order
//   2 weighted8 : the same with the weights a, b, g truncated to 8 fractional bits (a real sampler: the reference wgpu path on a GPU)
// each bit-exact against the shim in the same mode (tests/test_gpu_vs_ref.py).
template <int FILTER>
__device__ __forceinline__ float trilinear_combine(float t000, float t100, float t010, float t110, float t001, float t101, float t011, float t111, float a, float b, float c) {
    if (FILTER == 0) {
        const float c00 = mixf(t000, t100, a), c10 = mixf(t010, t110, a), c01 = mixf(t001, t101, a), c11 = mixf(t011, t111, a);
        return mixf(mixf(c00, c10, b), mixf(c01, c11, b), c);
    }
    if (FILTER == 2) { a = floorf(a * 256.0f) / 256.0f; b = floorf(b * 256.0f) / 256.0f; c = floorf(c * 256.0f) / 256.0f; }
    const float na = 1.0f - a, nb = 1.0f - b, nc = 1.0f - c;
    float acc = na * nb * nc * t000;
    acc = acc + a * nb * nc * t100;
    acc = acc + na * b * nc * t010;
    acc = acc + a * b * nc * t110;
    acc = acc + na * nb * c * t001;
    acc = acc + a * nb * c * t101;
    acc = acc + na * b * c * t011;
    acc = acc + a * b * c * t111;
    return acc;
}
template <int FILTER = 0, class Fetch>
__device__ __forceinline__ float trilinear_clamp(const Grid& g, Fetch fetch, float tx, float ty, float tz) {
    const float ux = tx * (float)g.nx - 0.5f, uy = ty * (float)g.ny - 0.5f, uz = tz * (float)g.nz - 0.5f;
    const float fx0 = floorf(ux), fy0 = floorf(uy), fz0 = floorf(uz);
    const float fx = ux - fx0, fy = uy - fy0, fz = uz - fz0;
    const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    const int xa = min(max(x0, 0), g.nx - 1), xb = min(max(x0 + 1, 0), g.nx - 1);
    const int ya = min(max(y0, 0), g.ny - 1), yb = min(max(y0 + 1, 0), g.ny - 1);
    const int za = min(max(z0, 0), g.nz - 1), zb = min(max(z0 + 1, 0), g.nz - 1);
    return trilinear_combine<FILTER>(fetch(xa, ya, za), fetch(xb, ya, za), fetch(xa, yb, za), fetch(xb, yb, za), fetch(xa, ya, zb), fetch(xb, ya, zb), fetch(xa, yb, zb), fetch(xb, yb, zb), fx, fy, fz);
}
// The same filter for an f32 volume with the x-pairs fetched as ONE 8-byte load each (4-byte aligned): the two x-texels of a
// pair are neighbours in memory, so the eight scattered 4-byte loads become four 8-byte ones (the particle kernels are
// issue-bound on the memory pipe: profiles/r01_pmc_sq_sparse_bench.csv, k_correct).  Same values, same arithmetic.
typedef float float2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float2_a4 ld_pair_o(const float* base, uint32_t cell) { return *reinterpret_cast<const float2_a4*>(reinterpret_cast<const char*>(base) + cell * 4u); }   // 32-bit offset from a uniform base (see ld4o)
template <int FILTER = 0>
__device__ __forceinline__ float trilinear_clamp_f32(const Grid& g, const float* __restrict__ V, float tx, float ty, float tz) {
    const float ux = tx * (float)g.nx - 0.5f, uy = ty * (float)g.ny - 0.5f, uz = tz * (float)g.nz - 0.5f;
    const float fx0 = floorf(ux), fy0 = floorf(uy), fz0 = floorf(uz);
    const float fx = ux - fx0, fy = uy - fy0, fz = uz - fz0;
    const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    const int xa = min(max(x0, 0), g.nx - 1), xb = min(max(x0 + 1, 0), g.nx - 1);
    const int ya = min(max(y0, 0), g.ny - 1), yb = min(max(y0 + 1, 0), g.ny - 1);
    const int za = min(max(z0, 0), g.nz - 1), zb = min(max(z0 + 1, 0), g.nz - 1);
    const int xbase = min(xa, g.nx - 2);                     // the pair (xbase, xbase + 1) always lies inside the row
    const bool lo_first = xa == xbase, hi_first = xb == xbase;
    auto pair = [&](int y, int z, float& lo, float& hi) {
        const float2_a4 q = ld_pair_o(V, (uint32_t)cidx(g, xbase, y, z));
        lo = lo_first ? q.x : q.y; hi = hi_first ? q.x : q.y;
    };
    float a0, a1, b0, b1, c0, c1, d0, d1;
    pair(ya, za, a0, a1); pair(yb, za, b0, b1); pair(ya, zb, c0, c1); pair(yb, zb, d0, d1);
    return trilinear_combine<FILTER>(a0, a1, b0, b1, c0, c1, d0, d1, fx, fy, fz);
}
// advect_particles.comp:139-148 / density_projection_correct_particles.comp:51-60 (Q12: literal)
__device__ __forceinline__ void truncate_step(const float* orig, const float* move, float* dir, float& max_step) {
    const float len = sqrtf((move[0] * move[0] + move[1] * move[1]) + move[2] * move[2]) + 1e-10f;
    float ms = len;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dir[k] = move[k] / len;
        const float pic = fractf(orig[k]);
        ms = fminf(ms, (dir[k] > 0.0f ? pic : 1.0f - pic) / fabsf(dir[k]) - 0.001f);
    }
    max_step = ms;
}

// =================================================================================================================
// A1: advect_particles.comp:35-194 (G2P + APIC rows + RK4-in-cell + wall handling + marker / density list)
// =================================================================================================================
// (4 waves per SIMD: the kernel holds 143 registers at the compiler's free choice = 3 waves per SIMD; capped at 128 it spills 7 of them and runs 11 % faster -- it is
//  bound by the latency of its 3 x 8 x 4 texel fetches: 32.5 -> 28.8 us per step on the headline scene, 329 -> 291 us at 10 M particles; 5 waves = 96 registers: slower)
template <int FILTER = 0>
__global__ __launch_bounds__(256, 4) void k_advect(Grid g, uint32_t num_particles, float dt, float4* __restrict__ pos, float4* __restrict__ pvx,
                                                float4* __restrict__ pvy, float4* __restrict__ pvz, const float* __restrict__ vx,
                                                const float* __restrict__ vy, const float* __restrict__ vz, const float4* __restrict__ solid,
                                                int8_t* __restrict__ marker, uint32_t* __restrict__ heads, uint8_t* __restrict__ brick_fluid,
                                                int nbx, int nby, const uint32_t* __restrict__ n_dev, uint32_t n_sel) {
    num_particles = particle_count(num_particles, n_dev, n_sel);
    if (blockIdx.x * 256u >= num_particles) return;      // (uniform)
    const uint32_t pi = blockIdx.x * 256 + threadIdx.x;
    const bool live = pi < num_particles;          // dead lanes run the (cheap) arithmetic on a dummy particle: the wave-level list insertion needs all lanes
    const float gs[3] = {(float)g.nx, (float)g.ny, (float)g.nz};
    const float inv[3] = {1.0f / gs[0], 1.0f / gs[1], 1.0f / gs[2]};
    const int dimm1[3] = {g.nx - 1, g.ny - 1, g.nz - 1};
    const float4 p0 = live ? pos[pi] : make_float4(1.5f, 1.5f, 1.5f, 0.0f);
    float op[3] = {p0.x, p0.y, p0.z};
    if (solid) {   // :46-65
        const float4 cs = solid_point_clamp(solid, g, op[0] * inv[0], op[1] * inv[1], op[2] * inv[2]);
        if (cs.w > 0.0f) {
            const float ax = fabsf(cs.x), ay = fabsf(cs.y), az = fabsf(cs.z);
            if (ax > ay) { if (ax > az) op[0] += signf(cs.x); else op[2] += signf(cs.z); }
            else { if (ay > az) op[1] += signf(cs.y); else op[2] += signf(cs.z); }
        }
    }
    float v[8][3], ipx[3], ipy[3], ipz[3];
    const float* vel[3] = {vx, vy, vz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // :74-93
        float o[3]; int lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { o[k] = fmaxf(0.0f, op[k] - (k == i ? 1.0f : 0.5f)); lo[k] = (int)o[k]; hi[k] = min(lo[k] + 1, dimm1[k]); }
        ipx[i] = fractf(o[0]); ipy[i] = fractf(o[1]); ipz[i] = fractf(o[2]);
        const float* V = vel[i];
        // the two x-texels of a pair are neighbours in memory: one 8-byte load per pair (see trilinear_clamp_f32).  Positions are
        // inside [0.001, dim - 0.001] here (clamped every step, the solid escape moves by one cell), so lo / hi are valid texels;
        // the clamps only keep wild caller-supplied positions from reading outside the volume.
        const int xbase = min(max(min(lo[0], g.nx - 2), 0), g.nx - 2);
        const bool lo_first = lo[0] <= xbase, hi_first = hi[0] <= xbase;
        auto pair = [&](int y, int z, float& a, float& b) {
            const float2_a4 q = ld_pair_o(V, (uint32_t)cidx(g, xbase, min(max(y, 0), g.ny - 1), min(max(z, 0), g.nz - 1)));
            a = lo_first ? q.x : q.y; b = hi_first ? q.x : q.y;
        };
        pair(lo[1], lo[2], v[0][i], v[1][i]); pair(hi[1], lo[2], v[2][i], v[3][i]);
        pair(lo[1], hi[2], v[4][i], v[5][i]); pair(hi[1], hi[2], v[6][i], v[7][i]);
    }
    float nv[3], cx[3], cy[3], cz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // :97-112
        const float x00 = mixf(v[0][i], v[1][i], ipx[i]), x01 = mixf(v[4][i], v[5][i], ipx[i]);
        const float x10 = mixf(v[2][i], v[3][i], ipx[i]), x11 = mixf(v[6][i], v[7][i], ipx[i]);
        const float xy0 = mixf(x00, x10, ipy[i]), xy1 = mixf(x01, x11, ipy[i]);
        nv[i] = mixf(xy0, xy1, ipz[i]);
        cx[i] = mixf(mixf(v[1][i], v[3][i], ipy[i]), mixf(v[5][i], v[7][i], ipy[i]), ipz[i]) -
                mixf(mixf(v[0][i], v[2][i], ipy[i]), mixf(v[4][i], v[6][i], ipy[i]), ipz[i]);
        cy[i] = mixf(x10, x11, ipz[i]) - mixf(x00, x01, ipz[i]);
        cz[i] = xy1 - xy0;
    }
    auto tri = [&](const float* sx, const float* sy, const float* sz, float* out) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            out[i] = mixf(mixf(mixf(v[0][i], v[1][i], sx[i]), mixf(v[2][i], v[3][i], sx[i]), sy[i]),
                          mixf(mixf(v[4][i], v[5][i], sx[i]), mixf(v[6][i], v[7][i], sx[i]), sy[i]), sz[i]);
    };
    float k2[3], k3[3], k4[3], st[3], sx[3], sy[3], sz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { st[i] = dt * 0.5f * nv[i]; sx[i] = satf(ipx[i] + st[i]); sy[i] = satf(ipy[i] + st[i]); sz[i] = satf(ipz[i] + st[i]); }   // :117-119 (Q11)
    tri(sx, sy, sz, k2);
#pragma unroll
    for (int i = 0; i < 3; ++i) { st[i] = dt * 0.5f * k2[i]; sx[i] = satf(ipx[i] + st[i]); sy[i] = satf(ipy[i] + st[i]); sz[i] = satf(ipz[i] + st[i]); }
    tri(sx, sy, sz, k3);
#pragma unroll
    for (int i = 0; i < 3; ++i) { st[i] = dt * k3[i]; sx[i] = satf(ipx[i] + st[i]); sy[i] = satf(ipy[i] + st[i]); sz[i] = satf(ipz[i] + st[i]); }
    tri(sx, sy, sz, k4);
    float mv[3], np[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { mv[i] = dt * (1.0f / 6.0f) * (nv[i] + 2.0f * (k2[i] + k3[i]) + k4[i]); np[i] = op[i] + mv[i]; }   // :126-127
    bool outside = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (clampf(np[k], 1.001f, gs[k] - 1.001f) != np[k]) outside = true;
    const float tc[3] = {np[0] * inv[0], np[1] * inv[1], np[2] * inv[2]};
    if (outside || (solid && solid_point_clamp(solid, g, tc[0], tc[1], tc[2]).w > 0.0f)) {   // :137-173
        float dir[3], ms;
        truncate_step(op, mv, dir, ms);
#pragma unroll
        for (int k = 0; k < 3; ++k) mv[k] = dir[k] * ms;
        if ((int)op[0] == (int)np[0] && (int)op[1] == (int)np[1] && (int)op[2] == (int)np[2]) {   // :154
            // push out of the solid along the gradient of the voxelisation's w channel; without solid voxels every sample is 0 and
            // the six trilinear evaluations (a few hundred instructions for every wave that touches a wall) are skipped
            if (solid) {
                auto sw = [&](int ax, int ay, int az) -> float { return solid[cidx(g, ax, ay, az)].w; };
                const float push[3] = {
                    trilinear_clamp<FILTER>(g, sw, tc[0] - inv[0], tc[1], tc[2]) - trilinear_clamp<FILTER>(g, sw, tc[0] + inv[0], tc[1], tc[2]),
                    trilinear_clamp<FILTER>(g, sw, tc[0], tc[1] - inv[1], tc[2]) - trilinear_clamp<FILTER>(g, sw, tc[0], tc[1] + inv[1], tc[2]),
                    trilinear_clamp<FILTER>(g, sw, tc[0], tc[1], tc[2] - inv[2]) - trilinear_clamp<FILTER>(g, sw, tc[0], tc[1], tc[2] + inv[2])};
#pragma unroll
                for (int k = 0; k < 3; ++k) mv[k] += push[k] * (dt * 50.0f);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) mv[k] += 0.0f * (dt * 50.0f);      // (0 - 0) * (dt * 50): keeps the sign-of-zero behaviour of the sum
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { np[k] = op[k] + mv[k]; np[k] = clampf(np[k], 1.001f, gs[k] - 1.001f); nv[k] = (dir[k] * ms) / dt; }
    }
    // :176-181 marker + density list (dual cell = ivec3(pos - 0.5)).  heads == nullptr: a z-slab group inserts the
    // particles after migration instead (blub_slab.hip.h: k_slab_insert_density_ghosts)
    uint32_t nxt = INVALID_LL;
    if (heads) {
        if (live) {
            const int x = (int)np[0], y = (int)np[1], z = (int)np[2];
            if (inb(g, x, y, z)) {
                const int c = cidx(g, x, y, z);
                const bool interior = !solid && x > 0 && y > 0 && z > 0 && x < g.nx - 1 && y < g.ny - 1 && z < g.nz - 1;   // (see k_build_lists)
                if (interior || marker[c] != CELL_SOLID) marker[c] = CELL_FLUID;
            }
        }
        const int dx = (int)(np[0] - 0.5f), dy = (int)(np[1] - 0.5f), dz = (int)(np[2] - 0.5f);
        // (rounds 1-2 grouped only runs of adjacent lanes here: the full group search cost 5 us more while it was a ds_bpermute per round; as a
        //  v_readlane loop it is as fast at 256^3 and 3 % faster with 8 M particles)
        nxt = wave_list_insert(heads, (live && inb(g, dx, dy, dz)) ? cidx(g, dx, dy, dz) : -1, pi);
    }
    if (!live) return;
    if (brick_fluid) {   // what k_bricks_mark_particles would do for the list build that follows (one launch less per step): 16 x 8 x 4 bricks
        const int x = (int)np[0], y = (int)np[1], z = (int)np[2];
        if (inb(g, x, y, z)) brick_fluid[((z >> 2) * nby + (y >> 3)) * nbx + (x >> 4)] = 1;
    }
    pos[pi] = make_float4(np[0], np[1], np[2], __uint_as_float(nxt));
    pvx[pi] = make_float4(cx[0], cx[1], cx[2], nv[0]);   // :186-188 (Q2: literal row layout)
    pvy[pi] = make_float4(cy[0], cy[1], cy[2], nv[1]);
    pvz[pi] = make_float4(cz[0], cz[1], cz[2], nv[2]);
}

// =================================================================================================================
// R3: density_projection_correct_particles.comp:25-73
// =================================================================================================================
// step_done_host / step_number: the run-ahead throttle of blub_fluid_step (a counter in pinned host memory; this is the last kernel of a
// step, and its last workgroup is dispatched when nearly all others have retired -- the throttle needs no more than that)
template <int FILTER = 0>
__global__ __launch_bounds__(256) void k_correct(Grid g, uint32_t num_particles, float4* __restrict__ pos, const int8_t* __restrict__ marker,
                                                 const float* __restrict__ vx, const float* __restrict__ vy, const float* __restrict__ vz,
                                                 volatile uint32_t* step_done_host, uint32_t step_number,
                                                 uint8_t* __restrict__ brick_fluid, int nbx, int nby, const uint32_t* __restrict__ n_dev, uint32_t n_sel) {
    if (step_done_host && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *step_done_host = step_number;
    num_particles = particle_count(num_particles, n_dev, n_sel);
    const uint32_t pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= num_particles) return;
    const float gs[3] = {(float)g.nx, (float)g.ny, (float)g.nz};
    const float inv[3] = {1.0f / gs[0], 1.0f / gs[1], 1.0f / gs[2]};
    const float4 p0 = pos[pi];
    const float op[3] = {p0.x, p0.y, p0.z};
    const float* vel[3] = {vx, vy, vz};
    float ch[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = fmaxf(0.0f, op[k] - (k == c ? 0.5f : 0.0f));
        const float* V = vel[c];
        ch[c] = trilinear_clamp_f32<FILTER>(g, V, o[0] * inv[0], o[1] * inv[1], o[2] * inv[2]);
    }
    float np[3] = {op[0] + ch[0], op[1] + ch[1], op[2] + ch[2]};
    bool outside = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (clampf(np[k], 1.001f, gs[k] - 1.001f) != np[k]) outside = true;
    bool in_solid = false;
    if (!outside) {
        const int x = min(max((int)floorf(np[0] * inv[0] * gs[0]), 0), g.nx - 1);
        const int y = min(max((int)floorf(np[1] * inv[1] * gs[1]), 0), g.ny - 1);
        const int z = min(max((int)floorf(np[2] * inv[2] * gs[2]), 0), g.nz - 1);
        in_solid = marker[cidx(g, x, y, z)] == CELL_SOLID;
    }
    if (outside || in_solid) {
        float dir[3], ms;
        truncate_step(op, ch, dir, ms);
#pragma unroll
        for (int k = 0; k < 3; ++k) { np[k] = op[k] + dir[k] * ms; np[k] = clampf(np[k], 1.001f, gs[k] - 1.001f); }
    }
    pos[pi] = make_float4(np[0], np[1], np[2], p0.w);
    if (brick_fluid) {   // what k_bricks_mark_particles would do for the next step's first list build (16 x 8 x 4 bricks, see k_advect)
        const int x = (int)np[0], y = (int)np[1], z = (int)np[2];
        if (inb(g, x, y, z)) brick_fluid[((z >> 2) * nby + (y >> 3)) * nbx + (x >> 4)] = 1;
    }
}

// =================================================================================================================
// B1-B3: particle_binning_{count,prefixsum,rewrite_particles}.comp (hybrid_fluid.rs:857-893), "fixed" semantics (Q4):
// guarded threads, 0-based destinations.  The scan is a deterministic three-phase wave-shuffle scan over the cell
// counters (block totals -> scan of totals -> rescan + offset) instead of the reference's block scan + one global
// atomic per block, so cells are laid out in linear-index order (a legal instance of the reference's "sloppy" order).
// =================================================================================================================
__global__ __launch_bounds__(256) void k_bin_count(Grid g, uint32_t num_particles, float4* __restrict__ pos, uint32_t* __restrict__ counters,
                                                   const uint32_t* __restrict__ n_dev, uint32_t n_sel) {
    num_particles = particle_count(num_particles, n_dev, n_sel);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_particles) return;
    const float4 p = pos[i];
    const int x = (int)p.x, y = (int)p.y, z = (int)p.z;
    uint32_t slot = 0;
    if (inb(g, x, y, z)) slot = atomicAdd(counters + cidx(g, x, y, z), 1u);
    reinterpret_cast<uint32_t*>(pos)[4 * (size_t)i + 3] = slot;                         // particle_binning_count.comp:12
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if (lane >= off) v += t; }
    return v;
}
constexpr int SCAN_ITEMS = 4;                 // u32 per thread (one uint4)
constexpr int SCAN_BLOCK = 1024 * SCAN_ITEMS; // cells per 1024-thread block
__global__ __launch_bounds__(1024) void k_scan_block_totals(const uint32_t* __restrict__ counters, int n, uint32_t* __restrict__ block_totals) {
    __shared__ uint32_t sm[16];
    const int base = (blockIdx.x * 1024 + threadIdx.x) * SCAN_ITEMS;
    uint32_t v = 0;
    if (base + 3 < n) { const uint4 c = *reinterpret_cast<const uint4*>(counters + base); v = c.x + c.y + c.z + c.w; }
    else for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) v += counters[base + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += sm[w]; block_totals[blockIdx.x] = t; }
}
// single block: exclusive scan of the block totals in place (nblocks <= 1024*64)
__global__ __launch_bounds__(1024) void k_scan_totals(uint32_t* __restrict__ block_totals, int nblocks) {
    __shared__ uint32_t sm[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int start = 0; start < nblocks; start += 1024) {
        const int i = start + threadIdx.x;
        const uint32_t v = i < nblocks ? block_totals[i] : 0u;
        uint32_t inc = wave_inclusive_scan(v);
        if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += sm[w];
        const uint32_t c = carry;
        if (i < nblocks) block_totals[i] = c + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + inc;
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_scan_apply(uint32_t* __restrict__ counters, int n, const uint32_t* __restrict__ block_offsets) {
    __shared__ uint32_t sm[16];
    const int base = (blockIdx.x * 1024 + threadIdx.x) * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS] = {0, 0, 0, 0};
    if (base + 3 < n) { const uint4 q = *reinterpret_cast<const uint4*>(counters + base); c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w; }
    else for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) c[k] = counters[base + k];
    const uint32_t tsum = c[0] + c[1] + c[2] + c[3];
    const uint32_t inc = wave_inclusive_scan(tsum);
    if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += sm[w];
    uint32_t run = off + inc - tsum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { run += c[k]; c[k] = run; }   // inclusive prefix (particle_binning_prefixsum.comp:35-57)
    if (base + 3 < n) *reinterpret_cast<uint4*>(counters + base) = make_uint4(c[0], c[1], c[2], c[3]);
    else for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) counters[base + k] = c[k];
}
// one_based: the literal Q4 reading (BLUB_BINNING_LITERAL) -- destination `inclusive - slot` as the shader writes it, slot 0 never written
__global__ __launch_bounds__(256) void k_bin_rewrite(Grid g, uint32_t num_particles, uint32_t max_particles, const float4* __restrict__ old_pos,
                                                     float4* __restrict__ new_pos, const uint32_t* __restrict__ inclusive, int one_based,
                                                     const uint32_t* __restrict__ n_dev, uint32_t n_sel) {
    num_particles = particle_count(num_particles, n_dev, n_sel);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_particles) return;
    const float4 p = old_pos[i];
    const int x = (int)p.x, y = (int)p.y, z = (int)p.z;
    const uint32_t inc = inb(g, x, y, z) ? inclusive[cidx(g, x, y, z)] : 0u;
    const uint32_t dst = inc - __float_as_uint(p.w) - (one_based ? 0u : 1u);            // particle_binning_rewrite_particles.comp:15; "fixed": 0-based (Q4)
    if (dst < max_particles) new_pos[dst] = p;                                          // (an out-of-bounds store is dropped)
}

// end-of-step marker for steps without particles (otherwise k_correct writes it)
__global__ void k_step_done(volatile uint32_t* host_counter, uint32_t step_number) { *host_counter = step_number; }

}  // namespace blubk
