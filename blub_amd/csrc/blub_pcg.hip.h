// Fused PCG iteration for the parity-default ("zero") preconditioner reading -- two streaming kernels per iteration.
//
// Reference schedule per iteration (pressure_solver.rs:654-723): apply_coeff, reduce x2, update p/r, [reduce_max x2],
// preconditioner pass 0, pass 1, reduce x2, update_search  (9-11 dispatches, 4N-byte reduce buffers).  Here:
//   KD "direction": [max|r| test of the previous iteration]  beta = sigma'/sigma;  s = M^-1 r + beta s  (own cell AND,
//                   redundantly, its 6 neighbours -- identical f32 ops, so every copy is bit-identical; s is double
//                   buffered because neighbours still need the old value);  partial s.As
//   KU "update"   : alpha = sigma / s.As;  p += alpha s;  r -= alpha A s;  partial max|r|;  partial (M^-1 r).r
// M^-1 r = (r/d)/d is pointwise (SURVEY Appendix B, Q1 reading "zero"), d = number of non-solid neighbours.
// All marker logic is folded once per solve into a stencil-descriptor byte volume `dvol`:
//   dvol[c] = 0x80 | d  for FLUID cells,  0 otherwise        (bit 7 = "is FLUID", bits 0-2 = diagonal of A)
// so the iteration kernels read 1 byte per cell + the f32 fields, never the marker.
// Dot products: one partial per block, re-reduced (<=1024 floats, L2 resident) by every block of the consumer kernel:
// no atomics, no reduce dispatch, bit-deterministic.  sigma is double-buffered by iteration parity.
//
// Two work mappings share the per-cell device functions:
//   *_z : dense, 2.5-D -- blub_pcg_dense.hip.h: tiles of T quads marched in z with register/LDS neighbour exchange
//                         (high fill ratios; the HBM-roofline path)
//   *_b : brick lists  -- here: 256-thread blocks, two FLUID bricks (16x8x4 cells) at a time, low fill ratios
//                         (the 1M @ 256^3 scene; latency- not byte-bound)
#pragma once
#include "blub_bricks.hip.h"

namespace blubk {

__device__ __forceinline__ int dbyte(uint32_t packed, int j) { return (int)((packed >> (8 * j)) & 0xFFu); }
__device__ __forceinline__ bool any_fluid_d(uint32_t packed) { return (packed & 0x80808080u) != 0u; }

struct QuadD { uint32_t c, ym, yp, zm, zp; int xm, xp; };
struct PcgTailSync { uint32_t arrivals; int timed_out; uint32_t pad[2]; };   // grid-barrier state of k_pcg_tail_b, zeroed by the init kernel of every solve
__device__ __forceinline__ void load_quad_d(const uint8_t* __restrict__ D, const Grid& g, int base, int x0, int y, int z, QuadD& q) {
    const int plane = g.nx * g.ny;
    q.xm = x0 > 0 ? (int)D[base - 1] : 0;
    q.xp = x0 + 4 < g.nx ? (int)D[base + 4] : 0;
    q.ym = y > 0 ? *reinterpret_cast<const uint32_t*>(D + base - g.nx) : 0u;
    q.yp = y + 1 < g.ny ? *reinterpret_cast<const uint32_t*>(D + base + g.nx) : 0u;
    q.zm = z > 0 ? *reinterpret_cast<const uint32_t*>(D + base - plane) : 0u;
    q.zp = z + 1 < g.nz ? *reinterpret_cast<const uint32_t*>(D + base + plane) : 0u;
}

template <int NT, bool MAX>
__device__ __forceinline__ float block_reduce(float v, float* sm) {
    v = MAX ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = sm[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = MAX ? fmaxf(r, sm[w]) : r + sm[w];
    return r;
}
template <int NT, bool MAX>
__device__ __forceinline__ float reduce_partials(const float* __restrict__ part, int n, float* sm) {
    float v = 0.0f;
    for (int i = threadIdx.x; i < n; i += NT) v = MAX ? fmaxf(v, part[i]) : v + part[i];
    return block_reduce<NT, MAX>(v, sm);
}

// ---- per-quad bodies ---------------------------------------------------------------------------------------------
// S0 (pressure_init.comp:19-84) + dvol + initial preconditioner/sigma (pressure_solver.rs:636-648)
__device__ __forceinline__ bool pcg_init_quad(const Grid& g, const int8_t* __restrict__ marker, uint8_t* __restrict__ dvol, float* __restrict__ p,
                                              float* __restrict__ r, float* __restrict__ s, int base, int x0, int y, int z, float& acc) {
    const uint32_t mc = *reinterpret_cast<const uint32_t*>(marker + base);
    float4 pc = ld4(p + base);
    uint32_t dq = 0;
    const bool anyf = any_fluid4(mc);
    if (anyf) {
        QuadMarkers m; m.c = mc; load_quad_markers(marker, g, base, x0, y, z, m);
        QuadValues pv; load_quad_values(p, g, base, x0, y, z, pv);
        const float4 rc = ld4(r + base);
        float4 so = ld4(s + base);
        float rr[4] = {rc.x, rc.y, rc.z, rc.w}, ss[4] = {so.x, so.y, so.z, so.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (mbyte(mc, j) != CELL_FLUID) continue;
            const int mX0 = j > 0 ? mbyte(m.c, j - 1) : m.xm, mX1 = j < 3 ? mbyte(m.c, j + 1) : m.xp;
            const int mY0 = mbyte(m.ym, j), mY1 = mbyte(m.yp, j), mZ0 = mbyte(m.zm, j), mZ1 = mbyte(m.zp, j);
            const int di = (mX0 != 0) + (mX1 != 0) + (mY0 != 0) + (mY1 != 0) + (mZ0 != 0) + (mZ1 != 0);
            const float d = (float)di;
            dq |= (uint32_t)(0x80 | di) << (8 * j);
            float res = rr[j];
            if (d > 0.0f) res -= d * f4(pv.c, j);                                   // :62-63
            if (mX0 == CELL_FLUID) res += (j > 0 ? f4(pv.c, j - 1) : pv.xm);          // :64-81
            if (mX1 == CELL_FLUID) res += (j < 3 ? f4(pv.c, j + 1) : pv.xp);
            if (mY0 == CELL_FLUID) res += f4(pv.ym, j);
            if (mY1 == CELL_FLUID) res += f4(pv.yp, j);
            if (mZ0 == CELL_FLUID) res += f4(pv.zm, j);
            if (mZ1 == CELL_FLUID) res += f4(pv.zp, j);
            rr[j] = res;
            ss[j] = precond_zero(res, d);
            acc += ss[j] * res;
        }
        *reinterpret_cast<float4*>(r + base) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4*>(s + base) = make_float4(ss[0], ss[1], ss[2], ss[3]);   // non-FLUID lanes keep their old value
    }
    *reinterpret_cast<uint32_t*>(dvol + base) = dq;
    bool dirty = false;   // pressure_init.comp:45-48: p := 0 outside the fluid
    if (mbyte(mc, 0) != CELL_FLUID && pc.x != 0.0f) { pc.x = 0.0f; dirty = true; }
    if (mbyte(mc, 1) != CELL_FLUID && pc.y != 0.0f) { pc.y = 0.0f; dirty = true; }
    if (mbyte(mc, 2) != CELL_FLUID && pc.z != 0.0f) { pc.z = 0.0f; dirty = true; }
    if (mbyte(mc, 3) != CELL_FLUID && pc.w != 0.0f) { pc.w = 0.0f; dirty = true; }
    if (dirty) *reinterpret_cast<float4*>(p + base) = pc;
    return anyf;
}

// A s for cell j from descriptor bytes (pressure.glsl:34-75: diag * s - sum over FLUID neighbours)
__device__ __forceinline__ float quad_mulA_d(const QuadD& m, const QuadValues& v, int j) {
    const int bX0 = j > 0 ? dbyte(m.c, j - 1) : m.xm, bX1 = j < 3 ? dbyte(m.c, j + 1) : m.xp;
    float r = 0.0f;
    r += (float)(dbyte(m.c, j) & 7) * f4(v.c, j);
    if (bX0 & 0x80) r -= (j > 0 ? f4(v.c, j - 1) : v.xm);
    if (bX1 & 0x80) r -= (j < 3 ? f4(v.c, j + 1) : v.xp);
    if (dbyte(m.ym, j) & 0x80) r -= f4(v.ym, j);
    if (dbyte(m.yp, j) & 0x80) r -= f4(v.yp, j);
    if (dbyte(m.zm, j) & 0x80) r -= f4(v.zm, j);
    if (dbyte(m.zp, j) & 0x80) r -= f4(v.zp, j);
    return r;
}
// FLUID-lane masks as bit operations (all ones / zero from bit 7 of descriptor byte j; v_bfi / v_and): the compiler turns `fluid ? a : b`
// around an LDS table read back into a branch per lane, which is exactly what the issue-bound kernels cannot afford
__device__ __forceinline__ uint32_t fluid_mask(uint32_t dq, int j) { return (uint32_t)((int)(dq << (24 - 8 * j)) >> 31); }
__device__ __forceinline__ float and_mask(float x, uint32_t m) { return __uint_as_float(__float_as_uint(x) & m); }
__device__ __forceinline__ float blend_mask(float a, float b, uint32_t m) { return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m)); }   // m ? a : b
// A u for cell j of a quad whose neighbourhood holds u = 0 on every non-FLUID cell (the caller guarantees it): the reference's "minus the
// FLUID neighbours" (pressure.glsl:34-75) needs no neighbour descriptors then -- subtracting a zero is exact -- which drops six LDS
// reads and the conditionals of quad_mulA_d from a kernel bound by its instruction stream.  Same operations in the same order.
__device__ __forceinline__ float quad_mulA_u(uint32_t dc, const QuadValues& v, int j) {
    float r = 0.0f;
    r += (float)(dbyte(dc, j) & 7) * f4(v.c, j);
    r -= (j > 0 ? f4(v.c, j - 1) : v.xm);
    r -= (j < 3 ? f4(v.c, j + 1) : v.xp);
    r -= f4(v.ym, j);
    r -= f4(v.yp, j);
    r -= f4(v.zm, j);
    r -= f4(v.zp, j);
    return r;
}
// 1/d for d = 0..7 in LDS (lut[0] = lut[1] = 1): the direction kernel needs 22 reciprocals per quad, and a table read is two
// instructions where the select chain of precond_zero_i is fifteen -- in a kernel whose run time is the latency of ONE wave's
// instruction stream.  The values are the same correctly rounded constants.
__device__ __forceinline__ void pcg_fill_inv_lut(float* lut) {
    if (threadIdx.x < 8) {
        const int di = (int)threadIdx.x;
        float inv = 1.0f;
        inv = di == 2 ? 0.5f : inv; inv = di == 3 ? (1.0f / 3.0f) : inv; inv = di == 4 ? 0.25f : inv; inv = di == 5 ? 0.2f : inv; inv = di >= 6 ? (1.0f / 6.0f) : inv;
        lut[di] = inv;
    }
}
__device__ __forceinline__ float snew_of(int dv, float r, float sold, float beta, const float* lut) {   // pressure_update_search.comp:23 on top of M^-1 r
    const float inv = lut[dv & 7];
    const float sn = (r * inv) * inv + beta * sold;   // evaluated unconditionally, masked by the FLUID bit (no select: see fluid_mask)
    return and_mask(sn, fluid_mask((uint32_t)dv, 0));
}
// (select-chain variants without a table: the dense 2.5-D kernels are bandwidth-, not issue-bound)
__device__ __forceinline__ float snew_of(int dv, float r, float sold, float beta) {
    const float sn = precond_zero_i(r, dv & 7) + beta * sold;
    return (dv & 0x80) ? sn : 0.0f;
}
__device__ __forceinline__ float4 snew4(uint32_t dq, const float4& r, const float4& s, float beta) {
    return make_float4(snew_of(dbyte(dq, 0), r.x, s.x, beta), snew_of(dbyte(dq, 1), r.y, s.y, beta),
                       snew_of(dbyte(dq, 2), r.z, s.z, beta), snew_of(dbyte(dq, 3), r.w, s.w, beta));
}
__device__ __forceinline__ float4 snew4(uint32_t dq, const float4& r, const float4& s, float beta, const float* lut) {
    return make_float4(snew_of(dbyte(dq, 0), r.x, s.x, beta, lut), snew_of(dbyte(dq, 1), r.y, s.y, beta, lut),
                       snew_of(dbyte(dq, 2), r.z, s.z, beta, lut), snew_of(dbyte(dq, 3), r.w, s.w, beta, lut));
}

// ---- load / compute split of the KD and KU bodies (brick mapping): all global loads of a quad are issued up front,
// unconditionally, so that they overlap the partial-reduction prologue; the sparse regime is latency- not byte-bound.
struct DirLoad { QuadD m; QuadValues sv, rv; int base, z; bool valid; };
template <bool FIRST>
__device__ __forceinline__ void dir_load(const Grid& g, const uint8_t* __restrict__ dvol, const float* __restrict__ r, const float* __restrict__ s_in,
                                         int base, int x0, int y, int z, DirLoad& L) {
    L.base = base; L.z = z;
    L.m.c = *reinterpret_cast<const uint32_t*>(dvol + base);
    load_quad_d(dvol, g, base, x0, y, z, L.m);
    load_quad_values(s_in, g, base, x0, y, z, L.sv);
    if (!FIRST) load_quad_values(r, g, base, x0, y, z, L.rv);
}
// HALO (z-slab groups): the quad also stores the s it computed for the ghost plane below `halo_lo` / above `halo_hi`
// (own planes of the slab, -1 = none), so the search direction needs no halo exchange of its own.
template <bool FIRST, bool HALO = false>
__device__ __forceinline__ void dir_compute(DirLoad& L, float* __restrict__ s_out, float beta, float& acc, const float* lut, int halo_lo = -1, int halo_hi = -1, int plane = 0) {
    if (!L.valid || !any_fluid_d(L.m.c)) return;
    QuadValues& sv = L.sv;
    const QuadD& m = L.m;
    if (!FIRST) {
        const QuadValues& rv = L.rv;
        const float4 sold = sv.c;
        sv.c = snew4(m.c, rv.c, sv.c, beta, lut);
        sv.ym = snew4(m.ym, rv.ym, sv.ym, beta, lut); sv.yp = snew4(m.yp, rv.yp, sv.yp, beta, lut);
        sv.zm = snew4(m.zm, rv.zm, sv.zm, beta, lut); sv.zp = snew4(m.zp, rv.zp, sv.zp, beta, lut);
        sv.xm = snew_of(m.xm, rv.xm, sv.xm, beta, lut); sv.xp = snew_of(m.xp, rv.xp, sv.xp, beta, lut);
        float4 so = sv.c;
        if (!(dbyte(m.c, 0) & 0x80)) so.x = sold.x;
        if (!(dbyte(m.c, 1) & 0x80)) so.y = sold.y;
        if (!(dbyte(m.c, 2) & 0x80)) so.z = sold.z;
        if (!(dbyte(m.c, 3) & 0x80)) so.w = sold.w;
        *reinterpret_cast<float4*>(s_out + L.base) = so;
        if (HALO) {
            if (L.z == halo_lo) *reinterpret_cast<float4*>(s_out + L.base - plane) = sv.zm;
            if (L.z == halo_hi) *reinterpret_cast<float4*>(s_out + L.base + plane) = sv.zp;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (dbyte(m.c, j) & 0x80) acc += f4(sv.c, j) * quad_mulA_d(m, sv, j);
}
struct UpdLoad { QuadD m; QuadValues sv; float4 pc, rc; int base; bool valid; };
__device__ __forceinline__ void upd_load(const Grid& g, const uint8_t* __restrict__ dvol, const float* __restrict__ s, const float* __restrict__ p,
                                         const float* __restrict__ r, int base, int x0, int y, int z, UpdLoad& L) {
    L.base = base;
    L.m.c = *reinterpret_cast<const uint32_t*>(dvol + base);
    load_quad_d(dvol, g, base, x0, y, z, L.m);
    load_quad_values(s, g, base, x0, y, z, L.sv);
    L.pc = ld4(p + base); L.rc = ld4(r + base);
}
__device__ __forceinline__ void upd_compute(const UpdLoad& L, float* __restrict__ p, float* __restrict__ r, float alpha, float& acc, float& emax) {
    if (!L.valid || !any_fluid_d(L.m.c)) return;
    float pp[4] = {L.pc.x, L.pc.y, L.pc.z, L.pc.w}, rr[4] = {L.rc.x, L.rc.y, L.rc.z, L.rc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dv = dbyte(L.m.c, j);
        if (!(dv & 0x80)) continue;
        const float as = quad_mulA_d(L.m, L.sv, j);
        pp[j] = pp[j] + alpha * f4(L.sv.c, j);
        float res = rr[j];
        res -= alpha * as;
        rr[j] = res;
        emax = fmaxf(emax, fabsf(res));
        acc += precond_zero_i(res, dv & 7) * res;
    }
    *reinterpret_cast<float4*>(p + L.base) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *reinterpret_cast<float4*>(r + L.base) = make_float4(rr[0], rr[1], rr[2], rr[3]);
}

// ---- shared prologues ----------------------------------------------------------------------------------------------
// Partials: the update kernel (and init) emit float2 {partial of (M^-1 r).r, partial of max|r|} per block, the
// direction kernel a float {partial of s.As}.  sigma_i is additionally stashed as a scalar by block 0 of KD(i), so each
// kernel re-reduces exactly ONE partial array.
template <int NT>
__device__ __forceinline__ float2 reduce_partials2(const float2* __restrict__ part, int n, float2* sm2) {
    float sx = 0.0f, mx = 0.0f;
    for (int i = threadIdx.x; i < n; i += NT) { const float2 p = part[i]; sx += p.x; mx = fmaxf(mx, p.y); }
    sx = wave_sum(sx); mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm2[wave] = make_float2(sx, mx);
    __syncthreads();
    float2 r = sm2[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) { r.x += sm2[w].x; r.y = fmaxf(r.y, sm2[w].y); }
    return r;
}
// KD prologue: sigma_i and the max|r| of iteration i-1 from the update partials, convergence test
// (pressure_reduce.comp:82-94), beta (RESULTMODE_BETA).  Returns false when the solve is finished; every block takes the
// same branch because the reductions are deterministic.
template <int NT>
__device__ __forceinline__ bool pcg_dir_prologue(PcgCtrl* __restrict__ ctrl, const float2* __restrict__ part_upd, int num_part, float tolerance,
                                                 int iteration, int check_prev, float2* sm2, float& beta) {
    if (ctrl->done) return false;     // uniform: finished solves cost one load, not a reduction
    const float sigma_prev = ctrl->sigma[(iteration + 1) & 1];
    const float2 red = reduce_partials2<NT>(part_upd, num_part, sm2);
    beta = 0.0f;
    if (iteration > 0) {
        if (check_prev && red.y < tolerance) {
            if (blockIdx.x == 0 && threadIdx.x == 0) { ctrl->max_err = red.y; ctrl->num_iter = (float)(iteration - 1); ctrl->done = 1; }
            return false;
        }
        beta = eps_div(red.x, sigma_prev);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->sigma[iteration & 1] = red.x;
    return true;
}
template <int NT>
__device__ __forceinline__ bool pcg_upd_prologue(const PcgCtrl* __restrict__ ctrl, const float* __restrict__ part_dir, int num_part, int iteration, float* sm, float& alpha) {
    if (ctrl->done) { alpha = 0.0f; return false; }
    const float sigma = ctrl->sigma[iteration & 1];
    const float sas = reduce_partials<NT, false>(part_dir, num_part, sm);
    alpha = eps_div(sigma, sas);                                                        // RESULTMODE_ALPHA
    return true;
}

// ---- brick-list wrappers ---------------------------------------------------------------------------------------------
// 256-thread blocks work on two bricks at a time (one per 128-thread half).  init runs over the ACTIVE list (dvol / p
// must be valid on every neighbour of a FLUID brick), KD / KU over the FLUID list.
#ifndef BLUB_PCG_BPB
#define BLUB_PCG_BPB 2
#endif
constexpr int PCG_BPB = BLUB_PCG_BPB;   // bricks per workgroup of the brick-mapped PCG kernels (one 128-thread slice per brick); measured on the 256^3 scene: 1 -> 645, 2 -> 773, 4 -> 762 steps/s
constexpr int PCG_B_THREADS = PCG_BPB * BRICK_THREADS;
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_init_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                              const int8_t* __restrict__ marker, uint8_t* __restrict__ dvol, float* __restrict__ p,
                                                              float* __restrict__ r, float* __restrict__ s, float2* __restrict__ part_upd, PcgCtrl* __restrict__ ctrl_to_clear,
                                                              PcgTailSync* __restrict__ sync_to_clear) {
    __shared__ float sm[8];
    if (ctrl_to_clear && blockIdx.x == 0 && threadIdx.x == 0) { PcgCtrl z{}; *ctrl_to_clear = z; }   // nobody reads it before the next kernel
    if (sync_to_clear && blockIdx.x == 0 && threadIdx.x == 0) { PcgTailSync z{}; *sync_to_clear = z; }
    float acc = 0.0f;
    const uint32_t n = *count;
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    for (uint32_t i = blockIdx.x * PCG_BPB + half; i < n; i += gridDim.x * PCG_BPB) {
        int x0, y, z;
        if (!brick_quad(bg, list[i], t, x0, y, z)) continue;
        (void)pcg_init_quad(bg.g, marker, dvol, p, r, s, cidx(bg.g, x0, y, z), x0, y, z, acc);
    }
    const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
    if (threadIdx.x == 0) part_upd[blockIdx.x] = make_float2(tot, 0.0f);
}
template <bool FIRST, bool HALO = false>
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_dir_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                             const uint8_t* __restrict__ dvol, const float* __restrict__ r, const float* __restrict__ s_in, float* __restrict__ s_out,
                                                             const float2* __restrict__ part_upd, float* __restrict__ part_dir, int num_part,
                                                             PcgCtrl* __restrict__ ctrl, float tolerance, int iteration, int check_prev, int halo_lo = -1, int halo_hi = -1) {
    __shared__ float sm[8];
    __shared__ float2 sm2[PCG_B_THREADS / 64 > 4 ? PCG_B_THREADS / 64 : 4];
    __shared__ float sInv[8];
    pcg_fill_inv_lut(sInv);   // (published by the barriers of the prologue's reduction)
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    uint32_t i = blockIdx.x * PCG_BPB + half;
    // the block's first list entry is requested together with the list length (list[] has an entry per brick of the grid, so the
    // read is always in bounds): one dependent round trip to memory less before the field loads can be issued
    const uint32_t b0 = i < (uint32_t)bg.nb ? list[i] : 0u;
    const uint32_t n = *count;
    DirLoad L; L.valid = false;
    if (i < n) { int x0, y, z; L.valid = brick_quad(bg, b0, t, x0, y, z); if (L.valid) dir_load<FIRST>(bg.g, dvol, r, s_in, cidx(bg.g, x0, y, z), x0, y, z, L); }
    float beta;
    if (!pcg_dir_prologue<PCG_B_THREADS>(ctrl, part_upd, num_part, tolerance, iteration, check_prev, sm2, beta)) return;
    float acc = 0.0f;
    const int plane = bg.g.nx * bg.g.ny;
    dir_compute<FIRST, HALO>(L, s_out, beta, acc, sInv, halo_lo, halo_hi, plane);
    for (i += gridDim.x * PCG_BPB; i < n; i += gridDim.x * PCG_BPB) {
        int x0, y, z;
        L.valid = brick_quad(bg, list[i], t, x0, y, z);
        if (L.valid) dir_load<FIRST>(bg.g, dvol, r, s_in, cidx(bg.g, x0, y, z), x0, y, z, L);
        dir_compute<FIRST, HALO>(L, s_out, beta, acc, sInv, halo_lo, halo_hi, plane);
    }
    const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
    if (threadIdx.x == 0) part_dir[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_update_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                const uint8_t* __restrict__ dvol, const float* __restrict__ s, float* __restrict__ p,
                                                                float* __restrict__ r, const float* __restrict__ part_dir, float2* __restrict__ part_upd, int num_part,
                                                                const PcgCtrl* __restrict__ ctrl, int iteration) {
    __shared__ float sm[8];
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    uint32_t i = blockIdx.x * PCG_BPB + half;
    const uint32_t b0 = i < (uint32_t)bg.nb ? list[i] : 0u;   // requested together with the list length, see k_pcg_dir_b
    const uint32_t n = *count;
    UpdLoad L; L.valid = false;
    if (i < n) { int x0, y, z; L.valid = brick_quad(bg, b0, t, x0, y, z); if (L.valid) upd_load(bg.g, dvol, s, p, r, cidx(bg.g, x0, y, z), x0, y, z, L); }
    float alpha;
    if (!pcg_upd_prologue<PCG_B_THREADS>(ctrl, part_dir, num_part, iteration, sm, alpha)) return;
    float acc = 0.0f, emax = 0.0f;
    upd_compute(L, p, r, alpha, acc, emax);
    for (i += gridDim.x * PCG_BPB; i < n; i += gridDim.x * PCG_BPB) {
        int x0, y, z;
        L.valid = brick_quad(bg, list[i], t, x0, y, z);
        if (L.valid) upd_load(bg.g, dvol, s, p, r, cidx(bg.g, x0, y, z), x0, y, z, L);
        upd_compute(L, p, r, alpha, acc, emax);
    }
    const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
    const float mx = block_reduce<PCG_B_THREADS, true>(emax, sm);
    if (threadIdx.x == 0) part_upd[blockIdx.x] = make_float2(tot, mx);
}

// ---- LDS-staged brick kernels ------------------------------------------------------------------------------------------
// k_pcg_dir_b / k_pcg_update_b fetch, per quad, the centre and six neighbours of every field straight from global memory
// (~200 B of L1/L2 traffic per quad for ~50 algorithmic bytes) and the direction kernel evaluates s = M^-1 r + beta s seven
// times per cell.  The staged variants load every cell of the brick's (16+2) x (8+2) x (4+2) face-halo tile ONCE (coalesced
// rows), evaluate the new search direction once per tile cell, keep tile + stencil descriptors in LDS (7.2 KB per brick) and
// form A s from there.  Measured: the shorter instruction stream is what pays -- +4 % steps/s on the 1 M particles @ 256^3
// scene (~1300 bricks); with ~10 k bricks (8 M particles @ 512^3) both variants take the same time (the cache traffic was
// not the bound after all), below ~500 bricks the extra barrier pair costs ~1 %.
// Thread <-> quad <-> partial mapping and the per-cell arithmetic are those of the _b kernels, so results are bit-identical
// (tests/test_gpu_fullsize.py::test_staged_brick_kernels_are_bit_identical_to_the_plain_brick_kernels).
constexpr int ST_ROW = 24;                                   // floats per LDS tile row: [3] = x-1 halo, [4..19] = brick, [20] = x+16 halo
constexpr int ST_ROWS = (BY + 2) * (BZ + 2);                 // 60 rows: (y-1 .. y+8) x (z-1 .. z+4)
struct alignas(16) StagedTile { float s[ST_ROWS * ST_ROW]; uint8_t d[ST_ROWS * ST_ROW]; };

// tile element e of the 240 interior quads / 120 halo scalars -> row, x offset inside the row, global coordinates
__device__ __forceinline__ bool st_row_needed(int row) {    // corner rows (y halo AND z halo) feed no 7-point stencil
    const int yy = row % (BY + 2), zz = row / (BY + 2);
    return !((yy == 0 || yy == BY + 1) && (zz == 0 || zz == BZ + 1));
}
// the stencil of quad t of the brick from the staged tile
__device__ __forceinline__ void st_read_quad(const StagedTile& T, int t, QuadD& m, QuadValues& sv) {
    const int q = t & 3, yy = (t >> 2) & 7, zz = t >> 5;
    const int o = ((zz + 1) * (BY + 2) + (yy + 1)) * ST_ROW + 4 + 4 * q;
    sv.c = *reinterpret_cast<const float4*>(T.s + o);
    sv.ym = *reinterpret_cast<const float4*>(T.s + o - ST_ROW); sv.yp = *reinterpret_cast<const float4*>(T.s + o + ST_ROW);
    sv.zm = *reinterpret_cast<const float4*>(T.s + o - (BY + 2) * ST_ROW); sv.zp = *reinterpret_cast<const float4*>(T.s + o + (BY + 2) * ST_ROW);
    sv.xm = T.s[o - 1]; sv.xp = T.s[o + 4];
    m.c = *reinterpret_cast<const uint32_t*>(T.d + o);
    m.ym = *reinterpret_cast<const uint32_t*>(T.d + o - ST_ROW); m.yp = *reinterpret_cast<const uint32_t*>(T.d + o + ST_ROW);
    m.zm = *reinterpret_cast<const uint32_t*>(T.d + o - (BY + 2) * ST_ROW); m.zp = *reinterpret_cast<const uint32_t*>(T.d + o + (BY + 2) * ST_ROW);
    m.xm = (int)T.d[o - 1]; m.xp = (int)T.d[o + 4];
}

template <bool FIRST, bool HALO = false>
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_dir_s(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                             const uint8_t* __restrict__ dvol, const float* __restrict__ r, const float* __restrict__ s_in, float* __restrict__ s_out,
                                                             const float2* __restrict__ part_upd, float* __restrict__ part_dir, int num_part,
                                                             PcgCtrl* __restrict__ ctrl, float tolerance, int iteration, int check_prev, int halo_lo = -1, int halo_hi = -1) {
    __shared__ float sm[8];
    __shared__ float2 sm2[PCG_B_THREADS / 64 > 4 ? PCG_B_THREADS / 64 : 4];
    __shared__ float sInv[8];
    __shared__ StagedTile tiles[PCG_BPB];
    pcg_fill_inv_lut(sInv);   // (published by the barriers of the prologue's reduction)
    const Grid g = bg.g;
    const uint32_t n = *count;
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    const int plane = g.nx * g.ny;
    float beta;
    if (!pcg_dir_prologue<PCG_B_THREADS>(ctrl, part_upd, num_part, tolerance, iteration, check_prev, sm2, beta)) return;
    StagedTile& T = tiles[half];
    float acc = 0.0f;
    for (uint32_t ib = blockIdx.x; ib * PCG_BPB < n; ib += gridDim.x) {       // uniform trip count for both halves: barriers inside
        const uint32_t i = ib * PCG_BPB + half;
        const bool have = i < n;
        const uint32_t b = have ? list[i] : 0u;
        int bxb, byb, bzb; brick_coords(bg, b, bxb, byb, bzb);
        const int x0b = bxb * BX, y0b = byb * BY, z0b = bzb * BZ;
        if (have) {
            // phase 1a: the 240 interior quads of the tile
            for (int e = t; e < ST_ROWS * 4; e += BRICK_THREADS) {
                const int row = e >> 2, q = e & 3;
                if (!st_row_needed(row)) continue;
                const int gy = y0b + row % (BY + 2) - 1, gz = z0b + row / (BY + 2) - 1, gx = x0b + 4 * q;
                float4 sn = make_float4(0.f, 0.f, 0.f, 0.f);
                uint32_t dq = 0;
                if ((unsigned)gy < (unsigned)g.ny && (unsigned)gz < (unsigned)g.nz && gx < g.nx) {
                    const int base = cidx(g, gx, gy, gz);
                    dq = *reinterpret_cast<const uint32_t*>(dvol + base);
                    const float4 so = ld4(s_in + base);
                    if (FIRST) sn = so;
                    else {
                        sn = snew4(dq, ld4(r + base), so, beta, sInv);
                        const bool own = gy >= y0b && gy < y0b + BY && gz >= z0b && gz < z0b + BZ;
                        const bool ghost = HALO && ((gz == halo_lo - 1 && z0b == halo_lo) || (gz == halo_hi + 1 && z0b + BZ - 1 == halo_hi)) && gy >= y0b && gy < y0b + BY;
                        if ((own && any_fluid_d(dq)) || ghost) {
                            float4 w = sn;                                                  // non-FLUID lanes of an own quad keep their old value
                            if (own) { if (!(dbyte(dq, 0) & 0x80)) w.x = so.x; if (!(dbyte(dq, 1) & 0x80)) w.y = so.y; if (!(dbyte(dq, 2) & 0x80)) w.z = so.z; if (!(dbyte(dq, 3) & 0x80)) w.w = so.w; }
                            *reinterpret_cast<float4*>(s_out + base) = w;
                        }
                    }
                }
                *reinterpret_cast<float4*>(T.s + row * ST_ROW + 4 + 4 * q) = sn;
                *reinterpret_cast<uint32_t*>(T.d + row * ST_ROW + 4 + 4 * q) = dq;
            }
            // phase 1b: the x-1 / x+16 halo cells of the 32 rows that have them (own y and z)
            for (int e = t; e < BY * BZ * 2; e += BRICK_THREADS) {
                const int side = e & 1, yy = (e >> 1) % BY, zz = (e >> 1) / BY;
                const int row = (zz + 1) * (BY + 2) + (yy + 1);
                const int gx = side ? x0b + BX : x0b - 1, gy = y0b + yy, gz = z0b + zz;
                float sn = 0.0f; int dv = 0;
                if ((unsigned)gx < (unsigned)g.nx && gy < g.ny && gz < g.nz) {
                    const int c = cidx(g, gx, gy, gz);
                    dv = (int)dvol[c];
                    sn = FIRST ? s_in[c] : snew_of(dv, r[c], s_in[c], beta, sInv);
                }
                T.s[row * ST_ROW + (side ? 20 : 3)] = sn;
                T.d[row * ST_ROW + (side ? 20 : 3)] = (uint8_t)dv;
            }
        }
        __syncthreads();
        if (have) {
            int x0, y, z;
            if (brick_quad(bg, b, t, x0, y, z)) {
                QuadD m; QuadValues sv;
                st_read_quad(T, t, m, sv);
                if (any_fluid_d(m.c)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (dbyte(m.c, j) & 0x80) acc += f4(sv.c, j) * quad_mulA_d(m, sv, j);
                }
            }
        }
        __syncthreads();   // the tile is rewritten for the next brick
    }
    (void)plane;
    const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
    if (threadIdx.x == 0) part_dir[blockIdx.x] = tot;
}

__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_update_s(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                                const uint8_t* __restrict__ dvol, const float* __restrict__ s, float* __restrict__ p,
                                                                float* __restrict__ r, const float* __restrict__ part_dir, float2* __restrict__ part_upd, int num_part,
                                                                const PcgCtrl* __restrict__ ctrl, int iteration) {
    __shared__ float sm[8];
    __shared__ StagedTile tiles[PCG_BPB];
    const Grid g = bg.g;
    const uint32_t n = *count;
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    float alpha;
    if (!pcg_upd_prologue<PCG_B_THREADS>(ctrl, part_dir, num_part, iteration, sm, alpha)) return;
    StagedTile& T = tiles[half];
    float acc = 0.0f, emax = 0.0f;
    for (uint32_t ib = blockIdx.x; ib * PCG_BPB < n; ib += gridDim.x) {
        const uint32_t i = ib * PCG_BPB + half;
        const bool have = i < n;
        const uint32_t b = have ? list[i] : 0u;
        int bxb, byb, bzb; brick_coords(bg, b, bxb, byb, bzb);
        const int x0b = bxb * BX, y0b = byb * BY, z0b = bzb * BZ;
        UpdLoad L; L.valid = false;
        if (have) {
            int x0, y, z;
            L.valid = brick_quad(bg, b, t, x0, y, z);
            if (L.valid) { L.base = cidx(g, x0, y, z); L.pc = ld4(p + L.base); L.rc = ld4(r + L.base); }   // own p, r: in flight across the staging
            for (int e = t; e < ST_ROWS * 4; e += BRICK_THREADS) {
                const int row = e >> 2, q = e & 3;
                if (!st_row_needed(row)) continue;
                const int gy = y0b + row % (BY + 2) - 1, gz = z0b + row / (BY + 2) - 1, gx = x0b + 4 * q;
                float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
                uint32_t dq = 0;
                if ((unsigned)gy < (unsigned)g.ny && (unsigned)gz < (unsigned)g.nz && gx < g.nx) {
                    const int base = cidx(g, gx, gy, gz);
                    dq = *reinterpret_cast<const uint32_t*>(dvol + base);
                    sv = ld4(s + base);
                }
                *reinterpret_cast<float4*>(T.s + row * ST_ROW + 4 + 4 * q) = sv;
                *reinterpret_cast<uint32_t*>(T.d + row * ST_ROW + 4 + 4 * q) = dq;
            }
            for (int e = t; e < BY * BZ * 2; e += BRICK_THREADS) {
                const int side = e & 1, yy = (e >> 1) % BY, zz = (e >> 1) / BY;
                const int row = (zz + 1) * (BY + 2) + (yy + 1);
                const int gx = side ? x0b + BX : x0b - 1, gy = y0b + yy, gz = z0b + zz;
                float sv = 0.0f; int dv = 0;
                if ((unsigned)gx < (unsigned)g.nx && gy < g.ny && gz < g.nz) { const int c = cidx(g, gx, gy, gz); dv = (int)dvol[c]; sv = s[c]; }
                T.s[row * ST_ROW + (side ? 20 : 3)] = sv;
                T.d[row * ST_ROW + (side ? 20 : 3)] = (uint8_t)dv;
            }
        }
        __syncthreads();
        if (L.valid) st_read_quad(T, t, L.m, L.sv);
        upd_compute(L, p, r, alpha, acc, emax);
        __syncthreads();
    }
    const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
    const float mx = block_reduce<PCG_B_THREADS, true>(emax, sm);
    if (threadIdx.x == 0) part_upd[blockIdx.x] = make_float2(tot, mx);
}

// ---- persistent tail of a brick-mapped solve -------------------------------------------------------------------------
// The host launches iterations [0, first_iteration) as separate kernels (their count comes from the iteration counts of
// the last few steps) and then this ONE kernel for iterations [first_iteration, max_iterations].  Normally the solve has
// converged by then and the kernel is a single no-op launch instead of 2 x (max - first) of them.  Otherwise it runs the
// remaining iterations itself: the same per-quad bodies, with an agent-scope grid barrier between the phases
// (cdna_hip_programming.md G16: stores -> __syncthreads -> lane-0 release fence -> relaxed agent atomic; relaxed poll ->
// acquire fence -> __syncthreads -> plain loads).  The grid is 256 blocks (one per CU, always co-resident) and every
// spin is bounded: on a timeout the kernel reports num_iter = -1 instead of hanging.
__device__ __forceinline__ bool grid_barrier(uint32_t* counter, uint32_t target, int* timed_out_flag) {
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22) || __hip_atomic_load(timed_out_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
        }
        if (!ok) __hip_atomic_store(timed_out_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_tail_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                              const uint8_t* __restrict__ dvol, float* r, float* s_even, float* s_odd, float* p,
                                                              float2* part_upd, float* part_dir, int num_part_in, PcgCtrl* ctrl, float tolerance,
                                                              int first_iteration, int max_iterations, int check_frequency, PcgTailSync* sync,
                                                              uint32_t seq, PcgCtrl* host_snapshot) {
    __shared__ float sm[8];
    __shared__ float2 sm2[PCG_B_THREADS / 64 > 4 ? PCG_B_THREADS / 64 : 4];
    __shared__ float sInv[8];
    pcg_fill_inv_lut(sInv);
    __syncthreads();
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    auto publish = [&]() {   // what k_pcg_finalize does
        ctrl->seq = seq;
        if (host_snapshot) { host_snapshot->max_err = ctrl->max_err; host_snapshot->num_iter = ctrl->num_iter; __threadfence_system(); host_snapshot->seq = seq; }
    };
    if (ctrl->done) { if (leader) publish(); return; }
    const uint32_t n = *count;
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    const uint32_t nblocks = gridDim.x;
    uint32_t barrier_no = 0;
    float sigma_prev = ctrl->sigma[(first_iteration + 1) & 1];
    int num_part = num_part_in;      // the first reduction reads what the LAUNCHED update kernel wrote
    for (int it = first_iteration; it <= max_iterations; ++it) {
        // ---- KD(it)
        const float2 red = reduce_partials2<PCG_B_THREADS>(part_upd, num_part, sm2);
        const int prev = it - 1;
        const bool check_prev = prev > 0 && check_frequency > 0 && prev % check_frequency == 0;
        if (it > 0 && check_prev && red.y < tolerance) {
            if (leader) { ctrl->max_err = red.y; ctrl->num_iter = (float)prev; ctrl->done = 1; publish(); }
            return;
        }
        const float beta = it > 0 ? eps_div(red.x, sigma_prev) : 0.0f;
        const float sigma = red.x;
        const float* s_in = ((it - 1) & 1) ? s_odd : s_even;
        float* s_out = (it & 1) ? s_odd : s_even;
        float acc = 0.0f;
        for (uint32_t i = blockIdx.x * PCG_BPB + half; i < n; i += nblocks * PCG_BPB) {
            int x0, y, z;
            DirLoad L;
            L.valid = brick_quad(bg, list[i], t, x0, y, z);
            if (L.valid) { if (it == 0) dir_load<true>(bg.g, dvol, r, s_even, cidx(bg.g, x0, y, z), x0, y, z, L); else dir_load<false>(bg.g, dvol, r, s_in, cidx(bg.g, x0, y, z), x0, y, z, L); }
            if (it == 0) dir_compute<true>(L, s_out, beta, acc, sInv); else dir_compute<false>(L, s_out, beta, acc, sInv);
        }
        const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
        if (threadIdx.x == 0) part_dir[blockIdx.x] = tot;
        if (!grid_barrier(&sync->arrivals, nblocks * ++barrier_no, &sync->timed_out)) { if (leader) { ctrl->num_iter = -1.0f; ctrl->done = 1; publish(); } return; }
        // ---- KU(it)
        const float sas = reduce_partials<PCG_B_THREADS, false>(part_dir, (int)nblocks, sm);
        const float alpha = eps_div(sigma, sas);
        float acc2 = 0.0f, emax = 0.0f;
        const float* s_cur = (it & 1) ? s_odd : s_even;
        for (uint32_t i = blockIdx.x * PCG_BPB + half; i < n; i += nblocks * PCG_BPB) {
            int x0, y, z;
            UpdLoad L;
            L.valid = brick_quad(bg, list[i], t, x0, y, z);
            if (L.valid) upd_load(bg.g, dvol, s_cur, p, r, cidx(bg.g, x0, y, z), x0, y, z, L);
            upd_compute(L, p, r, alpha, acc2, emax);
        }
        const float tot2 = block_reduce<PCG_B_THREADS, false>(acc2, sm);
        const float mx2 = block_reduce<PCG_B_THREADS, true>(emax, sm);
        __syncthreads();   // every thread has read the incoming partials (first pass: more entries than blocks) before they are overwritten
        if (threadIdx.x == 0) part_upd[blockIdx.x] = make_float2(tot2, mx2);
        if (!grid_barrier(&sync->arrivals, nblocks * ++barrier_no, &sync->timed_out)) { if (leader) { ctrl->num_iter = -1.0f; ctrl->done = 1; publish(); } return; }
        sigma_prev = sigma;
        num_part = (int)nblocks;
    }
    // i == max_num_iterations reached without convergence (pressure_reduce.comp:84)
    const float2 fin = reduce_partials2<PCG_B_THREADS>(part_upd, num_part, sm2);
    if (leader) { ctrl->max_err = fin.y; ctrl->num_iter = (float)max_iterations; ctrl->done = 1; publish(); }
}

// After the last update (i == max_num_iterations): statistics are written unconditionally if nothing converged before
// (pressure_reduce.comp:84: MaxNumSolverIterations == iterationIdx).
__global__ __launch_bounds__(256) void k_pcg_finalize(PcgCtrl* __restrict__ ctrl, const float2* __restrict__ part_upd, int num_part, int iteration, uint32_t seq,
                                                      PcgCtrl* __restrict__ host_snapshot) {
    __shared__ float2 sm2[PCG_B_THREADS / 64 > 4 ? PCG_B_THREADS / 64 : 4];
    const int done = ctrl->done;
    const float2 red = reduce_partials2<256>(part_upd, num_part, sm2);
    if (threadIdx.x == 0) {
        if (!done) { ctrl->max_err = red.y; ctrl->num_iter = (float)iteration; ctrl->done = 1; }
        ctrl->seq = seq;
        if (host_snapshot) {   // statistics read-back (pressure_solver.rs:176-191) written straight into the pinned ring: payload, fence, tag
            host_snapshot->max_err = ctrl->max_err; host_snapshot->num_iter = ctrl->num_iter;
            __threadfence_system();
            host_snapshot->seq = seq;
        }
    }
}
// end-of-step marker written straight into pinned host memory (run-ahead throttle, see blub_fluid_step)
__global__ void k_step_done(volatile uint32_t* host_counter, uint32_t step_number) { *host_counter = step_number; }
// LOD0 path: its own kernels already wrote the statistics; only the read-back tag is missing
__global__ void k_pcg_tag(PcgCtrl* __restrict__ ctrl, uint32_t seq) { ctrl->seq = seq; }

}  // namespace blubk
