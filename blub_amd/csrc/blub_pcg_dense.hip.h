// Dense-row PCG kernels, 2.5-D formulation for gfx950 (the HBM-roofline path: high fluid fill ratios).
//
// A tile = T consecutive quads (4 x-cells each) of one z-plane in memory order (T = 512: 8 rows of a 256-wide grid =
// 8 KiB contiguous per f32 volume), marched over `zc` planes by one T-thread block:
//   * z-neighbours never touch memory: every thread keeps its quad of planes z-1, z, z+1 in registers and rotates them;
//   * y- and x-neighbours of plane z are other threads' centre registers: exchanged through a double-buffered LDS row
//     buffer (one barrier per plane); only the two halo rows of the tile edge and the row-continuation cells come from
//     global memory (L2 hits when the neighbouring tile runs on the same XCD: tiles are handed out XCD-contiguously);
//   * the loads of plane z+2 (and of the next plane's p, r) are issued before plane z is computed (software prefetch).
// Every f32 field is therefore fetched ~(1 + 2/rows)(1 + 2/zc) times instead of 3-5 times.
// The direction kernel computes s_new = M^-1 r + beta s ONCE per cell per tile (plus the halo ring) instead of 5.5x.
// Arithmetic per cell is identical to the brick mapping (same device functions), only the work mapping differs.
#pragma once
#include "blub_pcg.hip.h"

namespace blubk {

struct PcgGeomZ {
    Grid g;
    int qpr, qpp;         // quads per row / per plane
    int T;                // threads (= quads) per tile
    int plane_tiles;      // ceil(qpp / T)
    int zc, z_chunks;
    int tiles;
    int alternate_march;  // odd z-chunks march downwards (xcd_tile_pairs): bit 0 in k_pcg_dir_z, bit 1 in k_pcg_update_z
    // the tile flags (k_pcg_init_z) are those of the UPDATE kernel's geometry; the direction kernel may march `flag_factor` of those chunks in one
    // tile (its z-halo planes are 2 / zc of what it reads, the update kernel has none on p and r): its flag is the OR of the chunks it covers
    int flag_factor, flag_chunks;
};
__device__ __forceinline__ bool tile_has_fluid(const PcgGeomZ& gz, const uint8_t* __restrict__ tile_flags, int pt, int zci) {
    if (gz.flag_factor == 1) return tile_flags[zci * gz.plane_tiles + pt] != 0;
    bool any = false;
    for (int j = 0; j < gz.flag_factor; ++j) { const int zf = zci * gz.flag_factor + j; if (zf < gz.flag_chunks) any = any || tile_flags[zf * gz.plane_tiles + pt] != 0; }
    return any;
}

// XCD-contiguous tile order: block b runs on XCD b % 8 (observed dispatch order, a speed hint only), so give XCD k the
// contiguous tile range [k*tiles/8, (k+1)*tiles/8): y-adjacent tiles then share an L2 for their halo rows.
__device__ __forceinline__ int xcd_tile(int i, int tiles) {
    const int per = (tiles + 7) >> 3;
    const int t = (i & 7) * per + (i >> 3);
    return t;   // may be >= tiles for the padded tail: callers skip those
}

// The same ranges per XCD, with the two z-chunks of a PAIR interleaved (tile j of chunk 2c, tile j of chunk 2c + 1, tile j + 1 of chunk 2c, ...) when
// the XCD's range is a whole number of chunk pairs: together with the alternating march direction of k_pcg_dir_z (even chunks upwards, odd chunks
// downwards) the two workgroups either side of a chunk interface reach it at the same time -- both at the start or both at the end of their
// march -- and on the same XCD, so the second one finds the planes it shares with its neighbour in that XCD's L2 instead of fetching them again.
__device__ __forceinline__ int xcd_tile_pairs(int i, const PcgGeomZ& gz) {
    const int per = (gz.tiles + 7) >> 3, two = 2 * gz.plane_tiles;
    if (per % two != 0 || gz.tiles % 8 != 0) return xcd_tile(i, gz.tiles);
    const int j = i >> 3, pair = j / two, jj = j - pair * two;
    return (i & 7) * per + pair * two + (jj & 1) * gz.plane_tiles + (jj >> 1);
}

// Streaming accesses that must not displace the halo rows other tiles are about to re-read from the L2 (p and r in KU have no
// halo: each value is touched exactly once per kernel): non-temporal loads / stores.
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld4s(const float* p) {
    if (NT) { const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return ld4(p);
}
template <bool NT> __device__ __forceinline__ float4 ld4so(const float* base, uint32_t byte_off) { return ld4s<NT>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off)); }
template <bool NT> __device__ __forceinline__ void st4s(float* p, const float4& a) {
    if (NT) { v4f_t v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(p)); }
    else *reinterpret_cast<float4*>(p) = a;
}

template <bool NT> __device__ __forceinline__ void st4so(float* base, uint32_t byte_off, const float4& a) { st4s<NT>(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off), a); }

__device__ __forceinline__ float4 sel4(uint32_t dq, const float4& a, const float4& fallback) {   // FLUID lanes take a, others fallback
    return make_float4(blend_mask(a.x, fallback.x, fluid_mask(dq, 0)), blend_mask(a.y, fallback.y, fluid_mask(dq, 1)),
                       blend_mask(a.z, fallback.z, fluid_mask(dq, 2)), blend_mask(a.w, fallback.w, fluid_mask(dq, 3)));
}
__device__ __forceinline__ float4 zero_outside_fluid(uint32_t dq, const float4& a) {
    return make_float4(and_mask(a.x, fluid_mask(dq, 0)), and_mask(a.y, fluid_mask(dq, 1)), and_mask(a.z, fluid_mask(dq, 2)), and_mask(a.w, fluid_mask(dq, 3)));
}

// ---- KU: p += alpha s; r -= alpha A s; partial (M^-1 r).r and max|r|  (pressure_update_pressure_and_residual.comp:23-59)
// (Round 3 also built this kernel in the formulation of k_pcg_dir_z below -- halo rows in the exchange buffer, raw loads two planes
//  ahead -- and measured it under deterministic placement: 61.1 vs 59.6 us at 256^3, 538-558 vs 539-558 us at 512^3.  This kernel moves 21 B
//  per cell of mixed read / write traffic at 6.0 TB/s already; more bytes in flight do not help it.  Removed again.)
template <int T, bool NT = false>
__global__ __launch_bounds__(T) void k_pcg_update_z(PcgGeomZ gz, const uint8_t* __restrict__ dvol, const float* __restrict__ s, float* __restrict__ p,
                                                    float* __restrict__ r, const float* __restrict__ part_dir, float2* __restrict__ part_upd, int num_part,
                                                    const uint8_t* __restrict__ tile_flags, const PcgCtrl* __restrict__ ctrl, int iteration) {
    __shared__ float sm[T / 64 + 1];
    __shared__ float4 ls[2][T];
    __shared__ DivConst div_lut[8];
    pcg_fill_div_lut(div_lut);       // (the prologue's barriers publish it)
    float alpha;
    if (!pcg_upd_prologue<T>(ctrl, part_dir, num_part, iteration, sm, alpha)) return;
    const Grid g = gz.g;
    const int t = threadIdx.x, plane = g.nx * g.ny, qpr = gz.qpr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float acc = 0.0f, emax = 0.0f;
    const int padded = ((gz.tiles + 7) >> 3) << 3;
    for (int it = blockIdx.x; it < padded; it += gridDim.x) {
        const int tile = xcd_tile_pairs(it, gz);
        if (tile >= gz.tiles || !tile_flags[tile]) continue;
        const int pt = tile % gz.plane_tiles, zci = tile / gz.plane_tiles;
        const int q = pt * T + t;
        const bool valid = q < gz.qpp;
        const int x0 = (q % qpr) << 2, y = q / qpr;
        const int z_begin = zci * gz.zc, z_end = min(z_begin + gz.zc, g.nz);
        // planes in MARCH order u = 0 .. n - 1: physical plane z0 + dz u -- odd z-chunks march downwards (xcd_tile_pairs); the stencil keeps its
        // physical orientation, only the order in which a thread adds its planes to the partials differs
        const int n = z_end - z_begin;
        const bool down = (zci & 1) != 0 && (gz.alternate_march & 2) != 0;
        const int z0 = down ? z_end - 1 : z_begin, dz = down ? -1 : 1;
        const int dplane = dz * plane;
        const int row_base = valid ? (y * g.nx + x0) : 0;
        const bool edge_lo = valid && (t < qpr) && y > 0, edge_hi = valid && (t >= T - qpr || q + qpr >= gz.qpp) && y + 1 < g.ny;
        const bool in_lo = t >= qpr, in_hi = (t + qpr < T) && (q + qpr < gz.qpp);
        const bool xm_glob = valid && x0 > 0 && t == 0, xp_glob = valid && x0 + 4 < g.nx && (t == T - 1);
        // march planes u - 1 (m), u (c), u + 1 (p) in registers; plane u + 2 and the next plane's p, r are in flight
        float4 s_m = zero4, s_c = zero4, s_p = zero4, s_n = zero4, pc = zero4, rc = zero4, pn = zero4, rn = zero4;
        uint32_t d_m = 0, d_c = 0, d_p = 0, d_n = 0;
        // tile-edge values of plane z that other tiles own (two rows + the row-continuation cells): fetched ONE PLANE AHEAD like
        // everything else -- loaded and consumed inside the same iteration they put a full memory latency on every plane's critical path
        struct Halo { float4 lo, hi; uint32_t dlo, dhi; float xm, xp; int dxm, dxp; };
        auto load_halo = [&](int base, bool cond) -> Halo {
            Halo h; h.lo = zero4; h.hi = zero4; h.dlo = 0; h.dhi = 0; h.xm = 0.f; h.xp = 0.f; h.dxm = 0; h.dxp = 0;
            if (cond) {
                if (edge_lo && !in_lo) { h.lo = ld4(s + base - g.nx); h.dlo = *reinterpret_cast<const uint32_t*>(dvol + base - g.nx); }
                if (edge_hi && !in_hi) { h.hi = ld4(s + base + g.nx); h.dhi = *reinterpret_cast<const uint32_t*>(dvol + base + g.nx); }
                if (xm_glob) { h.xm = s[base - 1]; h.dxm = dvol[base - 1]; }
                if (xp_glob) { h.xp = s[base + 4]; h.dxp = dvol[base + 4]; }
            }
            return h;
        };
        // (only the waves at the tile edge have such values: the others skip the loads, the zero fills and the plane rotation of them)
        const bool wave_halo = __ballot((edge_lo && !in_lo) || (edge_hi && !in_hi) || xm_glob || xp_glob) != 0ull;
        Halo hc = load_halo(0, false), hn = hc;
        auto exists = [&](int u) -> bool { const int zz = z0 + dz * u; return zz >= 0 && zz < g.nz; };
        if (valid) {
            const int b0 = z0 * plane + row_base;
            // s is only defined on FLUID cells (the reference never writes it elsewhere): every value is zeroed outside the fluid as it
            // arrives, so that the stencil below needs no per-neighbour tests (quad_mulA_u) -- these kernels are bound by VALU issue as
            // much as by bytes
            d_c = *reinterpret_cast<const uint32_t*>(dvol + b0); s_c = zero_outside_fluid(d_c, ld4(s + b0));
            if (exists(-1)) { d_m = *reinterpret_cast<const uint32_t*>(dvol + b0 - dplane); s_m = zero_outside_fluid(d_m, ld4(s + b0 - dplane)); }
            if (exists(1)) { d_p = *reinterpret_cast<const uint32_t*>(dvol + b0 + dplane); s_p = zero_outside_fluid(d_p, ld4(s + b0 + dplane)); }
            if (any_fluid_d(d_c)) { pc = ld4s<NT>(p + b0); rc = ld4s<NT>(r + b0); }
            if (wave_halo) hc = load_halo(b0, any_fluid_d(d_c));
        }
        for (int u = 0; u < n; ++u) {
            const int base = (z0 + dz * u) * plane + row_base;
            const int buf = u & 1;
            // issue the loads of the planes ahead: they are consumed after this plane's compute
            const uint32_t ub = (uint32_t)base;
            if (valid && exists(u + 2) && u + 1 < n) { s_n = ld4o(s, (uint32_t)(base + 2 * dplane) * 4u); d_n = ldu32o(dvol, (uint32_t)(base + 2 * dplane)); }
            else { s_n = zero4; d_n = 0; }      // (s_n is zeroed outside the fluid when it rotates in, below)
            const bool work_next = valid && u + 1 < n && any_fluid_d(d_p);
            if (work_next) { pn = ld4so<NT>(p, (uint32_t)(base + dplane) * 4u); rn = ld4so<NT>(r, (uint32_t)(base + dplane) * 4u); }
            if (wave_halo) hn = load_halo(base + dplane, work_next);
            const bool work = valid && any_fluid_d(d_c);
            ls[buf][t] = s_c;
            // LDS-only barrier: a __syncthreads() would first drain vmcnt, i.e. wait for the planes just requested (the whole point of
            // requesting them ahead); the exchange is double buffered by plane parity, so one barrier per plane suffices
            lds_barrier();
            if (work) {
                QuadValues sv;
                sv.c = s_c; sv.zm = down ? s_p : s_m; sv.zp = down ? s_m : s_p;      // physical z - 1 / z + 1
                if (y > 0) { if (in_lo) sv.ym = ls[buf][t - qpr]; else sv.ym = zero_outside_fluid(hc.dlo, hc.lo); } else sv.ym = zero4;
                if (y + 1 < g.ny) { if (in_hi) sv.yp = ls[buf][t + qpr]; else sv.yp = zero_outside_fluid(hc.dhi, hc.hi); } else sv.yp = zero4;
                if (x0 > 0) { if (t > 0) sv.xm = ls[buf][t - 1].w; else sv.xm = and_mask(hc.xm, fluid_mask((uint32_t)hc.dxm, 0)); } else sv.xm = 0.f;
                if (x0 + 4 < g.nx) { if (t < T - 1) sv.xp = ls[buf][t + 1].x; else sv.xp = and_mask(hc.xp, fluid_mask((uint32_t)hc.dxp, 0)); } else sv.xp = 0.f;
                float pp[4] = {pc.x, pc.y, pc.z, pc.w}, rr[4] = {rc.x, rc.y, rc.z, rc.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {                  // bit masks, no per-lane branches or selects
                    const uint32_t mk = fluid_mask(d_c, j);
                    const float as = quad_mulA_u(d_c, sv, j);
                    const float pj = pp[j] + alpha * f4(s_c, j);
                    float res = rr[j];
                    res -= alpha * as;
                    pp[j] = blend_mask(pj, pp[j], mk);
                    rr[j] = blend_mask(res, rr[j], mk);
                    const float zr = precond_exact(res, div_lut[dbyte(d_c, j) & 7]) * res;   // (M^-1 r) r with M^-1 r = (r / d) / d, correctly rounded
                    emax = fmaxf(emax, and_mask(fabsf(res), mk));
                    acc += and_mask(zr, mk);
                }
                st4so<NT>(p, ub * 4u, make_float4(pp[0], pp[1], pp[2], pp[3]));
                st4so<NT>(r, ub * 4u, make_float4(rr[0], rr[1], rr[2], rr[3]));
            }
            s_m = s_c; s_c = s_p; s_p = zero_outside_fluid(d_n, s_n); d_m = d_c; d_c = d_p; d_p = d_n; pc = pn; rc = rn;
            if (wave_halo) hc = hn;
        }
        __syncthreads();   // the LDS buffers are reused by the next tile
    }
    const float tot = block_reduce<T, false>(acc, sm);
    const float mx = block_reduce<T, true>(emax, sm);
    if (threadIdx.x == 0) part_upd[blockIdx.x] = make_float2(tot, mx);
}

// ---- KD: [convergence test] beta; s = M^-1 r + beta s; partial s.As  (pressure_update_search.comp + pressure_apply_coeff.comp)
// Round-3 formulation (the round-2 kernel issued 335 VALU instructions per quad and plane -- zero fills, 64-bit address arithmetic and a
// dozen exec-mask branches around the tile-edge cases -- and was bound by that, not by bytes: DESIGN.md 5d):
//   * the exchange buffer in LDS holds the tile's T quads of the current plane PLUS the qpr quads before and after them in memory order
//     (one row of halo either side; slot i <-> quad q0 - qpr + i).  Every thread then finds all four in-plane neighbours at FIXED
//     offsets from its own slot -- y-1 at [t], x-1 at [t + qpr - 1], x+1 at [t + qpr + 1], y+1 at [t + 2 qpr] -- with no case analysis:
//     row ends are two loop-invariant bit masks; the domain's y ends and a ragged last tile are zeros their writers put there;
//   * the 2 qpr halo quads are owned by the first 2 qpr threads (whole waves when qpr is a multiple of 64): they carry one extra quad's
//     raw loads through the same one-plane-ahead pipeline, convert it to s_new when it arrives and publish it next to their own value;
//   * loads are unconditional (lanes without a quad read the plane's first quad and mask the descriptor): no exec-mask branches around them;
//   * addresses: one 32-bit in-plane byte offset per thread (+ one per halo quad) against plane base pointers that advance in SGPRs;
//   * registers hold s_new of planes z-1, z, z+1; s_new of a plane is computed (and written to s_out) when the plane enters;
//   * s_new is 0 on every non-FLUID cell, so A s needs no neighbour descriptors (quad_mulA_u).  K(0) reads the stored s, which the
//     reference leaves untouched outside the fluid: it is zeroed outside the fluid as it arrives, which gives the same sums.
//   * the kernel is bound by the bytes it keeps IN FLIGHT, not by instruction issue (the first version of this formulation halved the
//     VALU count and got slower: one plane ahead = 36 B per thread at 4 waves per SIMD is ~46 KB per CU, Little's law wants that for
//     6 TB/s at 2 us).  Raw loads are therefore requested TWO planes ahead (own quad: plane z + 3 while plane z is computed; halo quad:
//     plane z + 2), in two register sets that alternate (the plane loop is unrolled by two), unconditionally and in a fixed order, so
//     that the compiler's s_waitcnt counts them exactly: planes that are not needed are read from the tile's first plane (cache hits)
//     with their descriptors masked.  Waves that own halo quads and waves that do not run two instantiations of the loop (HALO) for the
//     same reason -- a conditional load between the two sets would force vmcnt(0) at every use.
// Requires 2 qpr <= T (set_dense_geometry picks T accordingly; grids wider than 2048 cells use the brick mapping).
// Dynamic LDS: 2 x (T + 2 qpr) float4 (dense_dir_lds_bytes).
__host__ __device__ inline size_t dense_dir_lds_bytes(int T, int qpr) { return (size_t)2 * (size_t)(T + 2 * qpr) * 16u; }

struct DirRaw { uint32_t dq; float4 s, r; };
template <bool FIRST>
__device__ __forceinline__ void dir_raw_load(DirRaw& R, const uint8_t* __restrict__ dv, const float* __restrict__ sp, const float* __restrict__ rp, uint32_t off, uint32_t mask) {
    R.dq = ldu32o(dv, off >> 2) & mask;
    R.s = ld4o(sp, off);
    if (!FIRST) R.r = ld4o(rp, off); else R.r = make_float4(0.f, 0.f, 0.f, 0.f);
}
template <bool FIRST>
__device__ __forceinline__ float4 dir_snew(const DirRaw& R, float beta, const DivConst* lut) {      // s_new of a quad from its raw loads
    if (FIRST) return zero_outside_fluid(R.dq, R.s);
    return snew4(R.dq, R.r, R.s, beta, lut);
}
struct DirTile {      // per-thread constants of a tile
    uint32_t goff, vmask, mxm, mxp, hoff, hmask;
    int halo_slot, z_begin, z_end;
    bool valid, halo_thread;
    bool down;            // march from z_end - 1 to z_begin (odd z-chunks, see xcd_tile_pairs)
};
// the plane march of one tile; HALO: this wave owns halo quads; D: raw planes in flight per quad (own plane p and halo plane p live in
// register set p % D; the plane loop is unrolled by D so that the set indices are compile-time constants).  Planes are counted in MARCH
// order, u = 0 .. n - 1 (u = -1 and n: the z-halo planes): physical plane z0 + dz u, upwards for even z-chunks, downwards for odd ones.
// The stencil keeps its physical orientation (sv.zm is plane z - 1 whichever way the march runs), so A s of a cell does not depend on the
// direction; only the order in which a thread adds its planes to the s.As partial does.
template <int T, bool FIRST, bool NT, bool HALO, int D>
__device__ __forceinline__ void dir_march(const PcgGeomZ& gz, const DirTile& K, float4* __restrict__ ext, const uint8_t* __restrict__ dvol, const float* __restrict__ r,
                                          const float* __restrict__ s_in, float* __restrict__ s_out, float beta, const DivConst* lut, float& acc) {
    static_assert(D >= 2 && D <= 4, "pipeline depth");
    const Grid g = gz.g;
    const int t = threadIdx.x, qpr = gz.qpr, ext_n = T + 2 * qpr;
    const size_t plane = (size_t)g.nx * (size_t)g.ny;
    const int n = K.z_end - K.z_begin;
    const bool down = K.down;
    const int z0 = down ? K.z_end - 1 : K.z_begin, dz = down ? -1 : 1;
    // (uniform) base of march plane `u` if it exists and is wanted, else of the tile's first plane with a zero descriptor mask
    auto plane_of = [&](int u, bool wanted, uint32_t& pm) -> size_t { const int zz = z0 + dz * u; const bool ok = wanted && zz >= 0 && zz < g.nz; pm = ok ? 0xFFFFFFFFu : 0u; return (size_t)(ok ? zz : z0) * plane; };
    float4 n_m, n_c, n_p, h_c = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t d_c, d_p;
    DirRaw own[D], hal[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { hal[k].dq = 0; hal[k].s = hal[k].r = make_float4(0.f, 0.f, 0.f, 0.f); own[k].dq = 0; own[k].s = own[k].r = make_float4(0.f, 0.f, 0.f, 0.f); }
    {   // ---- planes -1, 0, 1 of the own quad and plane 0 of the halo quad: loaded and converted at once
        uint32_t pm_m, pm_c, pm_p;
        const size_t b_m = plane_of(-1, true, pm_m), b_c = plane_of(0, true, pm_c), b_p = plane_of(1, true, pm_p);
        DirRaw M, C, P, H;
        if (HALO) dir_raw_load<FIRST>(H, dvol + b_c, s_in + b_c, r + b_c, K.hoff, K.hmask);
        dir_raw_load<FIRST>(M, dvol + b_m, s_in + b_m, r + b_m, K.goff, K.vmask & pm_m);
        dir_raw_load<FIRST>(C, dvol + b_c, s_in + b_c, r + b_c, K.goff, K.vmask & pm_c);
        dir_raw_load<FIRST>(P, dvol + b_p, s_in + b_p, r + b_p, K.goff, K.vmask & pm_p);
        // ... and the first sets of the pipeline: halo planes 1 .. D - 1, own planes 2 .. D (in the order the loop consumes them)
#pragma unroll
        for (int j = 1; j < D; ++j) {
            uint32_t pm_h, pm_o;
            const size_t b_h = plane_of(j, j < n, pm_h), b_o = plane_of(j + 1, j < n, pm_o);
            if (HALO) dir_raw_load<FIRST>(hal[j % D], dvol + b_h, s_in + b_h, r + b_h, K.hoff, K.hmask & pm_h);
            dir_raw_load<FIRST>(own[(j + 1) % D], dvol + b_o, s_in + b_o, r + b_o, K.goff, K.vmask & pm_o);
        }
        n_m = dir_snew<FIRST>(M, beta, lut);                                      // z-halo plane: not written
        n_c = dir_snew<FIRST>(C, beta, lut);
        n_p = dir_snew<FIRST>(P, beta, lut);
        if (HALO) h_c = dir_snew<FIRST>(H, beta, lut);
        d_c = C.dq; d_p = P.dq;
        if (!FIRST && K.valid) {
            st4so<NT>(s_out + b_c, K.goff, n_c);
            if (1 < n) st4so<NT>(s_out + (size_t)(z0 + dz) * plane, K.goff, n_p);
        }
    }
    // one plane: request `issue` / `hissue` (own plane u + D + 1, halo plane u + D), stencil of plane u, then `use` / `huse` (own plane u + 2, halo plane u + 1) enter
    auto body = [&](int u, DirRaw& issue, DirRaw& hissue, DirRaw& use, DirRaw& huse) {
        uint32_t pm_o, pm_h;
        const size_t b_o = plane_of(u + D + 1, u + D < n, pm_o), b_h = plane_of(u + D, u + D < n, pm_h);
        if (HALO) dir_raw_load<FIRST>(hissue, dvol + b_h, s_in + b_h, r + b_h, K.hoff, K.hmask & pm_h);
        dir_raw_load<FIRST>(issue, dvol + b_o, s_in + b_o, r + b_o, K.goff, K.vmask & pm_o);
        // exchange of plane u (double buffered by plane parity: one LDS-only barrier per plane, see k_pcg_update_z)
        float4* const eb = ext + (u & 1) * ext_n;
        eb[t + qpr] = n_c;
        if (HALO) { if (K.halo_thread) eb[K.halo_slot] = h_c; }
        lds_barrier();
        {
            QuadValues sv;
            sv.c = n_c;
            sv.zm = down ? n_p : n_m; sv.zp = down ? n_m : n_p;                   // physical z - 1 / z + 1
            sv.ym = eb[t]; sv.yp = eb[t + 2 * qpr];
            sv.xm = and_mask(eb[t + qpr - 1].w, K.mxm); sv.xp = and_mask(eb[t + qpr + 1].x, K.mxp);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += f4(n_c, j) * quad_mulA_u(d_c, sv, j);      // n_c = 0 on non-FLUID lanes: they add an exact zero
        }
        const float4 n_n = dir_snew<FIRST>(use, beta, lut);
        if (HALO) h_c = dir_snew<FIRST>(huse, beta, lut);
        if (!FIRST && u + 2 < n && K.valid) st4so<NT>(s_out + (size_t)(z0 + dz * (u + 2)) * plane, K.goff, n_n);
        n_m = n_c; n_c = n_p; n_p = n_n; d_c = d_p; d_p = use.dq;
    };
    // at plane u = m D + k: own plane u + D + 1 goes into set (k + 1) % D, own plane u + 2 comes out of set (k + 2) % D; halo plane u + D into set k,
    // halo plane u + 1 out of set (k + 1) % D
    for (int u = 0; u < n; u += D) {
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (u + k < n) body(u + k, own[(k + 1) % D], hal[k % D], own[(k + 2) % D], hal[(k + 1) % D]);
    }
}

template <int T, bool FIRST, bool NT = false, int D = 2>
__global__ __launch_bounds__(T) void k_pcg_dir_z(PcgGeomZ gz, const uint8_t* __restrict__ dvol, const float* __restrict__ r, const float* __restrict__ s_in,
                                                 float* __restrict__ s_out, const float2* __restrict__ part_upd, float* __restrict__ part_dir, int num_part,
                                                 const uint8_t* __restrict__ tile_flags, PcgCtrl* __restrict__ ctrl, float tolerance, int iteration, int check_prev) {
    extern __shared__ float4 ext[];      // [2][T + 2 qpr]
    __shared__ float sm[T / 64 + 1];
    __shared__ float2 sm2[T / 64 + 1];
    __shared__ DivConst div_lut[8];
    pcg_fill_div_lut(div_lut);           // (the prologue's barriers publish it)
    float beta;
    if (!pcg_dir_prologue<T>(ctrl, part_upd, num_part, tolerance, iteration, check_prev, sm2, beta)) return;
    const Grid g = gz.g;
    const int t = threadIdx.x, qpr = gz.qpr;
    const bool halo_wave = (t & ~63) < 2 * qpr;            // wave-uniform
    float acc = 0.0f;
    const int padded = ((gz.tiles + 7) >> 3) << 3;
    for (int it = blockIdx.x; it < padded; it += gridDim.x) {
        const int tile = xcd_tile_pairs(it, gz);
        if (tile >= gz.tiles) continue;
        const int pt = tile % gz.plane_tiles, zci = tile / gz.plane_tiles;
        if (!tile_has_fluid(gz, tile_flags, pt, zci)) continue;
        const int q0 = pt * T, q = q0 + t;
        DirTile K;
        K.down = (zci & 1) != 0 && (gz.alternate_march & 1) != 0;
        K.valid = q < gz.qpp;
        const int x0 = (q % qpr) << 2;
        K.z_begin = zci * gz.zc; K.z_end = min(K.z_begin + gz.zc, g.nz);
        K.goff = K.valid ? (uint32_t)q * 16u : 0u;                                // byte offset of the own quad inside a plane of an f32 volume
        K.vmask = K.valid ? 0xFFFFFFFFu : 0u;
        K.mxm = (K.valid && x0 > 0) ? 0xFFFFFFFFu : 0u; K.mxp = (K.valid && x0 + 4 < g.nx) ? 0xFFFFFFFFu : 0u;
        // halo quad of this thread (threads 0 .. 2 qpr - 1): one row before the tile's first quad / one row after its last
        K.halo_thread = t < 2 * qpr;
        const int hq = t < qpr ? q0 - qpr + t : q0 + T + (t - qpr);
        const bool halo_valid = K.halo_thread && hq >= 0 && hq < gz.qpp;
        K.hoff = halo_valid ? (uint32_t)hq * 16u : 0u;
        K.hmask = halo_valid ? 0xFFFFFFFFu : 0u;
        K.halo_slot = t < qpr ? t : T + t;                                        // upper halo: T + qpr + (t - qpr)
        if (halo_wave) dir_march<T, FIRST, NT, true, D>(gz, K, ext, dvol, r, s_in, s_out, beta, div_lut, acc);
        else dir_march<T, FIRST, NT, false, D>(gz, K, ext, dvol, r, s_in, s_out, beta, div_lut, acc);
        __syncthreads();   // the LDS buffers are reused by the next tile
    }
    const float tot = block_reduce<T, false>(acc, sm);
    if (threadIdx.x == 0) part_dir[blockIdx.x] = tot;
}

// init for this mapping: same per-quad body as the row kernel, tile flags indexed by the z-march tiles
template <int T>
__global__ __launch_bounds__(T) void k_pcg_init_z(PcgGeomZ gz, const int8_t* __restrict__ marker, uint8_t* __restrict__ dvol, float* __restrict__ p,
                                                  float* __restrict__ r, float* __restrict__ s, float2* __restrict__ part_upd, uint8_t* __restrict__ tile_flags,
                                                  PcgCtrl* __restrict__ ctrl_to_clear) {
    __shared__ float sm[T / 64 + 1];
    if (ctrl_to_clear && blockIdx.x == 0 && threadIdx.x == 0) { PcgCtrl z{}; *ctrl_to_clear = z; }
    float acc = 0.0f;
    const Grid g = gz.g;
    const int padded = ((gz.tiles + 7) >> 3) << 3;
    for (int it = blockIdx.x; it < padded; it += gridDim.x) {
        const int tile = xcd_tile(it, gz.tiles);
        if (tile >= gz.tiles) continue;
        const int pt = tile % gz.plane_tiles, zci = tile / gz.plane_tiles;
        const int q = pt * T + threadIdx.x;
        const int x0 = (q % gz.qpr) << 2, y = q / gz.qpr;
        const int z_begin = zci * gz.zc, z_end = min(z_begin + gz.zc, g.nz);
        bool any = false;
        if (q < gz.qpp) for (int z = z_begin; z < z_end; ++z) any |= pcg_init_quad(g, marker, dvol, p, r, s, cidx(g, x0, y, z), x0, y, z, acc);
        const int tile_any = __syncthreads_or(any);
        if (threadIdx.x == 0) tile_flags[tile] = (uint8_t)(tile_any != 0);
    }
    const float tot = block_reduce<T, false>(acc, sm);
    if (threadIdx.x == 0) part_upd[blockIdx.x] = make_float2(tot, 0.0f);
}

}  // namespace blubk
