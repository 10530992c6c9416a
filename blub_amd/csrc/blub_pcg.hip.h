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
//   *_s : brick lists  -- here: 256-thread blocks, two FLUID bricks (16x8x4 cells) at a time, each brick's face-halo tile staged in
//                         LDS (low fill ratios: the 1M @ 256^3 scene; latency- not byte-bound)
// M^-1 r is evaluated EVERYWHERE as the reference writes it, (r / d) / d with two correctly rounded divisions (precond_exact below).
// Dot-product grouping of the brick mapping: see pcg_vblocks -- a function of the device-side list alone, not of the launch grid.
#pragma once
#include "blub_bricks.hip.h"

namespace blubk {

__device__ __forceinline__ int dbyte(uint32_t packed, int j) { return (int)((packed >> (8 * j)) & 0xFFu); }
__device__ __forceinline__ bool any_fluid_d(uint32_t packed) { return (packed & 0x80808080u) != 0u; }

struct QuadD { uint32_t c, ym, yp, zm, zp; int xm, xp; };
struct PcgTailSync { uint32_t arrivals; int timed_out; uint32_t pad[2]; };   // grid-barrier state of k_pcg1_tail_s, zeroed by the init kernel of every solve

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

// ---- M^-1 -----------------------------------------------------------------------------------------------------------
// M^-1 x exactly as the reference's two preconditioner passes write it with the Q1 reading "zero" (pressure_apply_preconditioner.comp:36-82):
// (x / d) / d with two correctly rounded divisions, d = number of non-SOLID neighbours in 0..6, skipped for d = 0.
// The iteration kernels are bound by instruction issue and evaluate this up to ~17 times per thread, so the IEEE division
// sequence (v_div_scale x2, v_rcp, 4-5 fma, v_div_fmas, v_div_fixup per division) is replaced by division by a CONSTANT:
// d = m 2^k with m in {1, 3, 5}; scaling by 2^-k is exact, and for the odd part
//     q0 = RN(y c), r = fma(-m, q0, y) (exact), q = fma(r, c, q0),  c = RN(1/m)
// is the correctly rounded y / m (Markstein's correction step; checked for every f32 significand by tests/native/div_const_check.c).
// The results are bit-identical to `(x / d) / d`.  (Round 1/2 multiplied by rounded reciprocals in the two-kernel schedule: gone.)
struct DivConst { float c, nm, sc2, pad; };     // per d = 0..7: RN(1/m), -m, 2^-2k  (both power-of-two scalings commute with the roundings: applied at once)
__device__ __forceinline__ void pcg_fill_div_lut(DivConst* lut) {   // by the first 8 threads of the block; a barrier must follow before the first use
    if (threadIdx.x < 8) {
        const int d = (int)threadIdx.x;
        DivConst e = {1.0f, -1.0f, 1.0f, 0.0f};                       // d = 0, 1 (and the impossible 7): the value itself
        if (d == 3 || d == 6) { e.c = 0x1.555556p-2f; e.nm = -3.0f; }  // RN(1/3)
        if (d == 5) { e.c = 0x1.99999ap-3f; e.nm = -5.0f; }            // RN(1/5)
        if (d == 2 || d == 6) e.sc2 = 0.25f;
        if (d == 4) e.sc2 = 0.0625f;
        lut[d] = e;
    }
}
__device__ __forceinline__ float precond_exact(float x, const DivConst& k) {
    const float y = x * k.sc2;
    float q = y * k.c;
    q = fmaf(fmaf(k.nm, q, y), k.c, q);                               // 2^-2k x / m
    float q2 = q * k.c;
    return fmaf(fmaf(k.nm, q2, q), k.c, q2);                          // ... / m  ==  (x / d) / d
}
// the same value by two IEEE divisions (once per solve: the init kernels)
__device__ __forceinline__ float precond_div(float x, float d) { if (d > 0.0f) { x /= d; x /= d; } return x; }

// ---- per-quad bodies ---------------------------------------------------------------------------------------------
// S0 (pressure_init.comp:19-84) + dvol + initial preconditioner/sigma (pressure_solver.rs:636-648)
// DIV: the right-hand side is the divergence of `dv`, formed here (D1, divergence_quad) instead of read from r
template <bool DIV = false>
__device__ __forceinline__ bool pcg_init_quad(const Grid& g, const int8_t* __restrict__ marker, uint8_t* __restrict__ dvol, float* __restrict__ p,
                                              float* __restrict__ r, float* __restrict__ s, int base, int x0, int y, int z, float& acc, const DivergenceSrc& dv = DivergenceSrc{}) {
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
        if (DIV) divergence_quad(g, m, dv, base, x0, y, z, rr);
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
            ss[j] = precond_div(res, d);
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
// s = M^-1 r + beta s (pressure_update_search.comp:23 on top of the two preconditioner passes), evaluated unconditionally and masked by the
// FLUID bit (no select: see fluid_mask); the divisor constants come from an 8-entry LDS table (one ds_read_b96 per cell)
__device__ __forceinline__ float snew_of(int dv, float r, float sold, float beta, const DivConst* lut) {
    const float sn = precond_exact(r, lut[dv & 7]) + beta * sold;
    return and_mask(sn, fluid_mask((uint32_t)dv, 0));
}
__device__ __forceinline__ float4 snew4(uint32_t dq, const float4& r, const float4& s, float beta, const DivConst* lut) {
    return make_float4(snew_of(dbyte(dq, 0), r.x, s.x, beta, lut), snew_of(dbyte(dq, 1), r.y, s.y, beta, lut),
                       snew_of(dbyte(dq, 2), r.z, s.z, beta, lut), snew_of(dbyte(dq, 3), r.w, s.w, beta, lut));
}

// KU body of one quad (pressure_update_pressure_and_residual.comp:23-59 + the z.r partial of the two preconditioner passes)
struct UpdLoad { QuadD m; QuadValues sv; float4 pc, rc; int base; bool valid; };
__device__ __forceinline__ void upd_compute(const UpdLoad& L, float* __restrict__ p, float* __restrict__ r, float alpha, float& acc, float& emax, const DivConst* lut) {
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
        acc += precond_exact(res, lut[dv & 7]) * res;
    }
    *reinterpret_cast<float4*>(p + L.base) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    *reinterpret_cast<float4*>(r + L.base) = make_float4(rr[0], rr[1], rr[2], rr[3]);
}

// ---- shared prologues ----------------------------------------------------------------------------------------------
// Partials: the update kernel (and init) emit float2 {partial of (M^-1 r).r, partial of max|r|} per (virtual) block, the
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
__device__ __forceinline__ bool pcg_dir_decide(PcgCtrl* __restrict__ ctrl, const float2 red, float sigma_prev, float tolerance, int iteration, int check_prev, float& beta) {
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
__device__ __forceinline__ bool pcg_dir_prologue(PcgCtrl* __restrict__ ctrl, const float2* __restrict__ part_upd, int num_part, float tolerance,
                                                 int iteration, int check_prev, float2* sm2, float& beta) {
    if (ctrl->done) return false;     // uniform: finished solves cost one load, not a reduction
    const float sigma_prev = ctrl->sigma[(iteration + 1) & 1];
    const float2 red = reduce_partials2<NT>(part_upd, num_part, sm2);
    return pcg_dir_decide(ctrl, red, sigma_prev, tolerance, iteration, check_prev, beta);
}
template <int NT>
__device__ __forceinline__ bool pcg_upd_prologue(const PcgCtrl* __restrict__ ctrl, const float* __restrict__ part_dir, int num_part, int iteration, float* sm, float& alpha) {
    if (ctrl->done) { alpha = 0.0f; return false; }
    const float sigma = ctrl->sigma[iteration & 1];
    const float sas = reduce_partials<NT, false>(part_dir, num_part, sm);
    alpha = eps_div(sigma, sas);                                                        // RESULTMODE_ALPHA
    return true;
}

// ---- brick-list kernels ------------------------------------------------------------------------------------------------
// 256-thread blocks work on two bricks at a time (one per 128-thread half).  init runs over the ACTIVE list (dvol / p
// must be valid on every neighbour of a FLUID brick), KD / KU over the FLUID list.
#ifndef BLUB_PCG_BPB
#define BLUB_PCG_BPB 2
#endif
constexpr int PCG_BPB = BLUB_PCG_BPB;   // bricks per workgroup of the brick-mapped PCG kernels (one 128-thread slice per brick); measured on the 256^3 scene: 1 -> 645, 2 -> 773, 4 -> 762 steps/s
constexpr int PCG_B_THREADS = PCG_BPB * BRICK_THREADS;

// VIRTUAL workgroups.  The brick lists are sparse work lists whose length the host only knows from a lagged, asynchronous snapshot, so the
// launch grid is an estimate.  The grouping of the dot-product partials must not depend on it (round-2 review: results moved with the host's
// timing): the work of a solve is cut into `pcg_vblocks(n)` virtual workgroups -- a function of the DEVICE-side list length alone --, virtual
// workgroup v sweeps the list entries v, v + V, v + 2V, ... (pairs of bricks) and owns partial slot v; a launched workgroup b executes the
// virtual workgroups b, b + gridDim.x, ...  Every consumer reduces the V partials in the same fixed order, so two solves on the same lists
// give bit-identical scalars whatever grid the host picked (normally grid >= V: one virtual workgroup per launched one, the surplus exits).
// V is a multiple of 8 (XCD-contiguous list order of the single-reduction kernels); `force` > 0: the z-slab groups pass the count every rank agreed on.
constexpr int PCG_VBLOCKS_MAX = 1024;
__host__ __device__ __forceinline__ int pcg_vblocks(uint32_t n_fluid, int force) {
    if (force > 0) return force;
    uint32_t v = (n_fluid + (uint32_t)PCG_BPB - 1u) / (uint32_t)PCG_BPB;
    v = v < 8u ? 8u : (v > (uint32_t)PCG_VBLOCKS_MAX ? (uint32_t)PCG_VBLOCKS_MAX : v);
    return (int)((v + 7u) & ~7u);
}

// The number of partials a consumer reduces (V) follows from the list length, which the kernel itself only learns from memory: loading the
// partials AFTER that would add a dependent round trip to kernels that last as long as their chain of round trips.  So each
// thread requests its share of the partial array together with the list length, bounded by the launch grid's own estimate of V (`spec`),
// and the estimate is corrected once the length has arrived: entries beyond V are dropped, entries the estimate missed (rare: the host's
// snapshot lagged behind a growing list) are fetched then.  Entries from PCG_VBLOCKS_MAX on (only the gathered arrays of z-slab groups,
// whose sizes are kernel arguments) are read by a plain loop.
constexpr int PCG_PART_PER_THREAD = PCG_VBLOCKS_MAX / PCG_B_THREADS;
template <class P> struct SpecPartials { P v[PCG_PART_PER_THREAD]; };
__device__ __forceinline__ void zero_of(float& x) { x = 0.0f; }
__device__ __forceinline__ void zero_of(float2& x) { x = make_float2(0.f, 0.f); }
__device__ __forceinline__ void zero_of(float4& x) { x = make_float4(0.f, 0.f, 0.f, 0.f); }
template <class P>
__device__ __forceinline__ void spec_partials_load(const P* __restrict__ part, int spec, SpecPartials<P>& L) {
#pragma unroll
    for (int k = 0; k < PCG_PART_PER_THREAD; ++k) {
        const int i = (int)threadIdx.x + k * PCG_B_THREADS;
        zero_of(L.v[k]);
        if (i < spec) L.v[k] = part[i];
    }
}
template <class P>
__device__ __forceinline__ void spec_partials_fix(const P* __restrict__ part, int spec, int num_part, SpecPartials<P>& L) {
#pragma unroll
    for (int k = 0; k < PCG_PART_PER_THREAD; ++k) {
        const int i = (int)threadIdx.x + k * PCG_B_THREADS;
        if (i >= num_part) zero_of(L.v[k]);
        else if (i >= spec) L.v[k] = part[i];
    }
}
__device__ __forceinline__ int spec_bound(int num_part_in) { return min(num_part_in > 0 ? num_part_in : (int)gridDim.x, PCG_VBLOCKS_MAX); }
__device__ __forceinline__ float2 reduce_spec2(const SpecPartials<float2>& L, const float2* __restrict__ part, int num_part, float2* sm2) {
    float sx = 0.0f, mx = 0.0f;
#pragma unroll
    for (int k = 0; k < PCG_PART_PER_THREAD; ++k) { sx += L.v[k].x; mx = fmaxf(mx, L.v[k].y); }
    for (int i = (int)threadIdx.x + PCG_VBLOCKS_MAX; i < num_part; i += PCG_B_THREADS) { const float2 p = part[i]; sx += p.x; mx = fmaxf(mx, p.y); }
    sx = wave_sum(sx); mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm2[wave] = make_float2(sx, mx);
    __syncthreads();
    float2 r = sm2[0];
#pragma unroll
    for (int w = 1; w < PCG_B_THREADS / 64; ++w) { r.x += sm2[w].x; r.y = fmaxf(r.y, sm2[w].y); }
    return r;
}
__device__ __forceinline__ float reduce_spec1(const SpecPartials<float>& L, const float* __restrict__ part, int num_part, float* sm) {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < PCG_PART_PER_THREAD; ++k) v += L.v[k];
    for (int i = (int)threadIdx.x + PCG_VBLOCKS_MAX; i < num_part; i += PCG_B_THREADS) v += part[i];
    return block_reduce<PCG_B_THREADS, false>(v, sm);
}

template <bool DIV>
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_init_b(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, const uint32_t* __restrict__ count_fluid,
                                                              int vb_force, const int8_t* __restrict__ marker, uint8_t* __restrict__ dvol, float* __restrict__ p,
                                                              float* __restrict__ r, float* __restrict__ s, float2* __restrict__ part_upd, PcgCtrl* __restrict__ ctrl_to_clear,
                                                              PcgTailSync* __restrict__ sync_to_clear, DivergenceSrc dv) {
    __shared__ float sm[8];
    if (ctrl_to_clear && blockIdx.x == 0 && threadIdx.x == 0) { PcgCtrl z{}; *ctrl_to_clear = z; }   // nobody reads it before the next kernel
    if (sync_to_clear && blockIdx.x == 0 && threadIdx.x == 0) { PcgTailSync z{}; *sync_to_clear = z; }
    const uint32_t n = *count;
    const int V = pcg_vblocks(*count_fluid, vb_force);
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    for (int vb = blockIdx.x; vb < V; vb += gridDim.x) {
        float acc = 0.0f;
        for (uint32_t i = (uint32_t)vb * PCG_BPB + half; i < n; i += (uint32_t)V * PCG_BPB) {
            int x0, y, z;
            if (!brick_quad(bg, list[i], t, x0, y, z)) continue;
            (void)pcg_init_quad<DIV>(bg.g, marker, dvol, p, r, s, cidx(bg.g, x0, y, z), x0, y, z, acc, dv);
        }
        const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
        if (threadIdx.x == 0) part_upd[vb] = make_float2(tot, 0.0f);
    }
}

// ---- LDS-staged brick tiles ------------------------------------------------------------------------------------------
// Every cell of the brick's (16+2) x (8+2) x (4+2) face-halo tile is loaded ONCE (coalesced rows), the new search direction is evaluated
// once per tile cell, tile + stencil descriptors stay in LDS (7.2 KB per brick) and A s is formed from there.  (Rounds 1-2 also carried
// plain variants that fetched centre + six neighbours of every field per quad straight from global memory and evaluated
// s = M^-1 r + beta s seven times per cell: bit-identical, 1 % faster below ~500 bricks, 4 % slower above -- removed.)
constexpr int ST_ROW = 24;                                   // floats per LDS tile row: [3] = x-1 halo, [4..19] = brick, [20] = x+16 halo
constexpr int ST_ROWS = (BY + 2) * (BZ + 2);                 // 60 rows: (y-1 .. y+8) x (z-1 .. z+4)
struct alignas(16) StagedTile { float s[ST_ROWS * ST_ROW]; uint8_t d[ST_ROWS * ST_ROW]; };

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
// Fills the staged tile of brick b: `quad(base, inside, own, gy, gz, dq_out)` returns the float4 of an interior quad of the tile (global cell
// index `base`, valid when `inside`; `own`: the quad belongs to the brick itself), `cell(c, inside, dv_out)` the value of an x-1 / x+16 halo cell.
template <class QuadFn, class CellFn>
__device__ __forceinline__ void st_fill(StagedTile& T, const BrickGeom& bg, uint32_t b, int t, QuadFn quad, CellFn cell) {
    const Grid g = bg.g;
    int bxb, byb, bzb; brick_coords(bg, b, bxb, byb, bzb);
    const int x0b = bxb * BX, y0b = byb * BY, z0b = bzb * BZ;
    for (int e = t; e < ST_ROWS * 4; e += BRICK_THREADS) {          // the 240 interior quads of the tile
        const int row = e >> 2, q = e & 3;
        if (!st_row_needed(row)) continue;
        const int gy = y0b + row % (BY + 2) - 1, gz = z0b + row / (BY + 2) - 1, gx = x0b + 4 * q;
        const bool inside = (unsigned)gy < (unsigned)g.ny && (unsigned)gz < (unsigned)g.nz && gx < g.nx;
        const bool own = gy >= y0b && gy < y0b + BY && gz >= z0b && gz < z0b + BZ;
        uint32_t dq = 0;
        const float4 v = quad(inside ? cidx(g, gx, gy, gz) : 0, inside, own, gy, gz, dq);
        *reinterpret_cast<float4*>(T.s + row * ST_ROW + 4 + 4 * q) = v;
        *reinterpret_cast<uint32_t*>(T.d + row * ST_ROW + 4 + 4 * q) = dq;
    }
    for (int e = t; e < BY * BZ * 2; e += BRICK_THREADS) {          // the x-1 / x+16 halo cells of the 32 rows that have them (own y and z)
        const int side = e & 1, yy = (e >> 1) % BY, zz = (e >> 1) / BY;
        const int row = (zz + 1) * (BY + 2) + (yy + 1);
        const int gx = side ? x0b + BX : x0b - 1, gy = y0b + yy, gz = z0b + zz;
        const bool inside = (unsigned)gx < (unsigned)g.nx && gy < g.ny && gz < g.nz;
        int dv = 0;
        const float v = cell(inside ? cidx(g, gx, gy, gz) : 0, inside, dv);
        T.s[row * ST_ROW + (side ? 20 : 3)] = v;
        T.d[row * ST_ROW + (side ? 20 : 3)] = (uint8_t)dv;
    }
}

// KD on the brick lists.  HALO (z-slab groups): the block also stores the s it computed for the ghost plane below `halo_lo` / above
// `halo_hi` (own planes of the slab, -1 = none), so the search direction needs no halo exchange of its own.
// num_part_in > 0: that many partials are reduced (z-slab groups: the gathered segments of all slabs); 0: the solve's own V.
// (the body as a device function: k_pcg_dir_s runs it once per launch, k_pcg_tail_s in a loop with grid barriers.  Returns false when the solve is
//  finished -- `done` was set, or this iteration's convergence test succeeded -- and nothing was computed.  SURPLUS_EXITS: launched workgroups
//  beyond the virtual ones return at once; the tail kernel's must stay, they take part in its grid barriers)
struct PcgBrickShared { float sm[8]; float2 sm2[PCG_B_THREADS / 64 > 4 ? PCG_B_THREADS / 64 : 4]; DivConst sDiv[8]; StagedTile tiles[PCG_BPB]; };
template <bool FIRST, bool HALO, bool SURPLUS_EXITS>
__device__ __forceinline__ bool pcg_dir_iteration(PcgBrickShared& S, BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int vb_force,
                                                  const uint8_t* __restrict__ dvol, const float* __restrict__ r, const float* __restrict__ s_in, float* __restrict__ s_out,
                                                  const float2* __restrict__ part_upd, float* __restrict__ part_dir, int num_part_in,
                                                  PcgCtrl* __restrict__ ctrl, float tolerance, int iteration, int check_prev, int halo_lo, int halo_hi) {
    float* const sm = S.sm; float2* const sm2 = S.sm2; DivConst* const sDiv = S.sDiv; StagedTile* const tiles = S.tiles;
    // one round trip: list length, `done`, sigma_{i-1} and this thread's share of the partials (spec_partials_load)
    const uint32_t n = *count;
    const int done = ctrl->done;
    const float sigma_prev = ctrl->sigma[(iteration + 1) & 1];
    const int spec = spec_bound(num_part_in);
    SpecPartials<float2> SP;
    spec_partials_load(part_upd, spec, SP);
    const int V = pcg_vblocks(n, vb_force);
    if ((SURPLUS_EXITS && (int)blockIdx.x >= V) || done) return false;
    const int num_part = num_part_in > 0 ? num_part_in : V;
    spec_partials_fix(part_upd, spec, num_part, SP);
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    float beta;
    if (!pcg_dir_decide(ctrl, reduce_spec2(SP, part_upd, num_part, sm2), sigma_prev, tolerance, iteration, check_prev, beta)) return false;
    StagedTile& T = tiles[half];
    for (int vb = blockIdx.x; vb < V; vb += gridDim.x) {
        float acc = 0.0f;
        for (uint32_t ib = (uint32_t)vb; ib * PCG_BPB < n; ib += (uint32_t)V) {       // uniform trip count for both halves: barriers inside
            const uint32_t i = ib * PCG_BPB + half;
            const bool have = i < n;
            const uint32_t b = have ? list[i] : 0u;
            if (have) {
                int bxb, byb, bzb; brick_coords(bg, b, bxb, byb, bzb); (void)bxb;
                const int y0b = byb * BY, z0b = bzb * BZ;
                st_fill(T, bg, b, t,
                    [&](int base, bool inside, bool own, int gy, int gz, uint32_t& dq) -> float4 {
                        if (!inside) return make_float4(0.f, 0.f, 0.f, 0.f);
                        dq = *reinterpret_cast<const uint32_t*>(dvol + base);
                        const float4 so = ld4(s_in + base);
                        if (FIRST) return so;
                        const float4 sn = snew4(dq, ld4(r + base), so, beta, sDiv);
                        const bool ghost = HALO && ((gz == halo_lo - 1 && z0b == halo_lo) || (gz == halo_hi + 1 && z0b + BZ - 1 == halo_hi)) && gy >= y0b && gy < y0b + BY;
                        if ((own && any_fluid_d(dq)) || ghost) {
                            float4 w = sn;                                                  // non-FLUID lanes of an own quad keep their old value
                            if (own) { if (!(dbyte(dq, 0) & 0x80)) w.x = so.x; if (!(dbyte(dq, 1) & 0x80)) w.y = so.y; if (!(dbyte(dq, 2) & 0x80)) w.z = so.z; if (!(dbyte(dq, 3) & 0x80)) w.w = so.w; }
                            *reinterpret_cast<float4*>(s_out + base) = w;
                        }
                        return sn;
                    },
                    [&](int c, bool inside, int& dv) -> float {
                        if (!inside) return 0.0f;
                        dv = (int)dvol[c];
                        return FIRST ? s_in[c] : snew_of(dv, r[c], s_in[c], beta, sDiv);
                    });
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
        const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
        if (threadIdx.x == 0) part_dir[vb] = tot;
    }
    return true;
}
template <bool FIRST, bool HALO = false>
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_dir_s(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int vb_force,
                                                             const uint8_t* __restrict__ dvol, const float* __restrict__ r, const float* __restrict__ s_in, float* __restrict__ s_out,
                                                             const float2* __restrict__ part_upd, float* __restrict__ part_dir, int num_part_in,
                                                             PcgCtrl* __restrict__ ctrl, float tolerance, int iteration, int check_prev, int halo_lo = -1, int halo_hi = -1) {
    __shared__ PcgBrickShared S;
    pcg_fill_div_lut(S.sDiv);   // (published by the barriers of the prologue's reduction)
    (void)pcg_dir_iteration<FIRST, HALO, true>(S, bg, list, count, vb_force, dvol, r, s_in, s_out, part_upd, part_dir, num_part_in, ctrl, tolerance, iteration, check_prev, halo_lo, halo_hi);
}

template <bool SURPLUS_EXITS>
__device__ __forceinline__ bool pcg_update_iteration(PcgBrickShared& S, BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int vb_force,
                                                     const uint8_t* __restrict__ dvol, const float* __restrict__ s, float* __restrict__ p,
                                                     float* __restrict__ r, const float* __restrict__ part_dir, float2* __restrict__ part_upd, int num_part_in,
                                                     const PcgCtrl* __restrict__ ctrl, int iteration) {
    float* const sm = S.sm; DivConst* const sDiv = S.sDiv; StagedTile* const tiles = S.tiles;
    const uint32_t n = *count;       // one round trip: see k_pcg_dir_s
    const int done = ctrl->done;
    const float sigma = ctrl->sigma[iteration & 1];
    const int spec = spec_bound(num_part_in);
    SpecPartials<float> SP;
    spec_partials_load(part_dir, spec, SP);
    const int V = pcg_vblocks(n, vb_force);
    if ((SURPLUS_EXITS && (int)blockIdx.x >= V) || done) return false;
    const int num_part = num_part_in > 0 ? num_part_in : V;
    spec_partials_fix(part_dir, spec, num_part, SP);
    const int t = threadIdx.x & (BRICK_THREADS - 1), half = threadIdx.x >> 7;
    const float alpha = eps_div(sigma, reduce_spec1(SP, part_dir, num_part, sm));      // RESULTMODE_ALPHA
    StagedTile& T = tiles[half];
    for (int vb = blockIdx.x; vb < V; vb += gridDim.x) {
        float acc = 0.0f, emax = 0.0f;
        for (uint32_t ib = (uint32_t)vb; ib * PCG_BPB < n; ib += (uint32_t)V) {
            const uint32_t i = ib * PCG_BPB + half;
            const bool have = i < n;
            const uint32_t b = have ? list[i] : 0u;
            UpdLoad L; L.valid = false;
            if (have) {
                int x0, y, z;
                L.valid = brick_quad(bg, b, t, x0, y, z);
                if (L.valid) { L.base = cidx(bg.g, x0, y, z); L.pc = ld4(p + L.base); L.rc = ld4(r + L.base); }   // own p, r: in flight across the staging
                st_fill(T, bg, b, t,
                    [&](int base, bool inside, bool, int, int, uint32_t& dq) -> float4 {
                        if (!inside) return make_float4(0.f, 0.f, 0.f, 0.f);
                        dq = *reinterpret_cast<const uint32_t*>(dvol + base);
                        return ld4(s + base);
                    },
                    [&](int c, bool inside, int& dv) -> float { if (!inside) return 0.0f; dv = (int)dvol[c]; return s[c]; });
            }
            __syncthreads();
            if (L.valid) st_read_quad(T, t, L.m, L.sv);
            upd_compute(L, p, r, alpha, acc, emax, sDiv);
            __syncthreads();
        }
        const float tot = block_reduce<PCG_B_THREADS, false>(acc, sm);
        const float mx = block_reduce<PCG_B_THREADS, true>(emax, sm);
        if (threadIdx.x == 0) part_upd[vb] = make_float2(tot, mx);
    }
    return true;
}
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_update_s(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int vb_force,
                                                                const uint8_t* __restrict__ dvol, const float* __restrict__ s, float* __restrict__ p,
                                                                float* __restrict__ r, const float* __restrict__ part_dir, float2* __restrict__ part_upd, int num_part_in,
                                                                const PcgCtrl* __restrict__ ctrl, int iteration) {
    __shared__ PcgBrickShared S;
    pcg_fill_div_lut(S.sDiv);   // (published by the barriers of the prologue's reduction)
    (void)pcg_update_iteration<true>(S, bg, list, count, vb_force, dvol, s, p, r, part_dir, part_upd, num_part_in, ctrl, iteration);
}

// ---- grid barrier of the persistent tail kernel (blub_pcg1.hip.h) -----------------------------------------------------
// cdna_hip_programming.md G16: stores -> __syncthreads -> lane-0 release fence -> relaxed agent atomic; relaxed poll ->
// acquire fence -> __syncthreads -> plain loads.  Every block must be co-resident (the host bounds the grid by the
// occupancy of the kernel) and every spin is bounded: on a timeout the kernel reports num_iter = -1 instead of hanging.
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

// Persistent tail of a solve in the reference's two-reduction order: the host launches the iteration pairs the last few solves needed (+ one
// check interval) and this ONE kernel for the rest.  Normally KD(first) finds the solve converged (or `done` already set), publishes the
// statistics -- k_pcg_finalize's job -- and the kernel is a single short launch instead of 2 x (max - first) of them; otherwise it runs the
// remaining iterations itself: the same bodies, the same virtual workgroups (bit-identical to the launched solve), two bounded grid barriers per
// iteration (one workgroup per CU, all co-resident).  Round 2 had this for the plain brick kernels; round 3 dropped it with them and a
// finished solve of the library's DEFAULT schedule then cost ~24 pairs of no-op launches (134 us per step on the headline scene).
__global__ __launch_bounds__(PCG_B_THREADS) void k_pcg_tail_s(BrickGeom bg, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, const uint8_t* __restrict__ dvol,
                                                              float* r, float* s_even, float* s_odd, float* p, float2* part_upd, float* part_dir, PcgCtrl* ctrl, float tolerance,
                                                              int first_iteration, int max_iterations, int check_frequency, PcgTailSync* sync, uint32_t seq, PcgCtrl* host_snapshot) {
    __shared__ PcgBrickShared S;
    __shared__ float2 smf[4];
    pcg_fill_div_lut(S.sDiv);
    __syncthreads();
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    auto publish = [&]() {   // what k_pcg_finalize does
        ctrl->seq = seq;
        if (host_snapshot) { host_snapshot->max_err = ctrl->max_err; host_snapshot->num_iter = ctrl->num_iter; __threadfence_system(); host_snapshot->seq = seq; }
    };
    if (ctrl->done) { if (leader) publish(); return; }      // uniform
    // (a barrier of THIS solve already gave up -- or the test hook "pcg_tail_inject_timeout" says so: the solve is reported unfinished)
    if (__hip_atomic_load(&sync->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { if (leader) { ctrl->num_iter = -1.0f; ctrl->done = 1; publish(); } return; }
    uint32_t barrier_no = 0;
    for (int it = first_iteration; it <= max_iterations; ++it) {
        const int prev = it - 1;
        const int check_prev = prev > 0 && check_frequency > 0 && prev % check_frequency == 0;
        const float* s_in = ((it - 1) & 1) ? s_odd : s_even;
        float* s_out = (it & 1) ? s_odd : s_even;
        const bool ran = pcg_dir_iteration<false, false, false>(S, bg, list, count, 0, dvol, r, s_in, s_out, part_upd, part_dir, 0, ctrl, tolerance, it, check_prev, -1, -1);
        if (!ran) { if (leader) publish(); return; }        // converged at the check of iteration it - 1 (the leader wrote the statistics itself)
        if (!grid_barrier(&sync->arrivals, gridDim.x * ++barrier_no, &sync->timed_out)) { if (leader) { ctrl->num_iter = -1.0f; ctrl->done = 1; publish(); } return; }
        (void)pcg_update_iteration<false>(S, bg, list, count, 0, dvol, s_out, p, r, part_dir, part_upd, 0, ctrl, it);
        if (!grid_barrier(&sync->arrivals, gridDim.x * ++barrier_no, &sync->timed_out)) { if (leader) { ctrl->num_iter = -1.0f; ctrl->done = 1; publish(); } return; }
    }
    // i == max_num_iterations reached without convergence (pressure_reduce.comp:84)
    if (blockIdx.x == 0) {
        const float2 fin = reduce_partials2<PCG_B_THREADS>(part_upd, pcg_vblocks(*count, 0), smf);
        if (threadIdx.x == 0) { ctrl->max_err = fin.y; ctrl->num_iter = (float)max_iterations; ctrl->done = 1; publish(); }
    }
}

// After the last update (i == max_num_iterations): statistics are written unconditionally if nothing converged before
// (pressure_reduce.comp:84: MaxNumSolverIterations == iterationIdx).  num_part > 0: that many partials; 0: the V of the brick-mapped solve.
__global__ __launch_bounds__(256) void k_pcg_finalize(PcgCtrl* __restrict__ ctrl, const float2* __restrict__ part_upd, int num_part, const uint32_t* __restrict__ count_fluid,
                                                      int iteration, uint32_t seq, PcgCtrl* __restrict__ host_snapshot) {
    __shared__ float2 sm2[4];
    const int done = ctrl->done;
    const float2 red = reduce_partials2<256>(part_upd, num_part > 0 ? num_part : pcg_vblocks(*count_fluid, 0), sm2);
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
// LOD0 path: its own kernels already wrote the statistics; only the read-back tag is missing
__global__ void k_pcg_tag(PcgCtrl* __restrict__ ctrl, uint32_t seq) { ctrl->seq = seq; }

}  // namespace blubk
