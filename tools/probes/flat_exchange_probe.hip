// Probe 3 (round 6): the synchronisation + exchange part of ONE iteration of a would-be persistent brick-mapped PCG WITHOUT a barrier: every piece of data another
// workgroup reads carries the iteration number in its own 16 bytes (the recipe of the slab groups' direct transport: tagged partials), so a consumer simply polls
// the data until the tag is current -- two dependent cross-XCD hops (the producer's write-through store, the consumer's cache-bypassing load) instead of the four
// of the two-level tree (tree_barrier_probe.hip: 7.1 us for 256 workgroups).
//   * every workgroup publishes `spb` 16-byte partials {gamma, delta, max|r|, tag} (spb virtual workgroups per resident one) and re-reduces ALL of them;
//   * every workgroup publishes `halo` 16-byte units {3 values, tag} and polls the units of its two list neighbours (what a brick's face cells would be);
//   * slots are double buffered by iteration parity (nobody can be two iterations ahead of anybody: an iteration needs everyone's partials);
//   * private work: a `priv` KB slice per workgroup read and rewritten with plain accesses (stands for the LDS-resident fields: set 0 for pure exchange).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/flat_exchange_probe.hip -o build/flat_exchange_probe && build/flat_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_sc1(const float4* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_sc1(float4* p, float4 a) {
    v4f v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void ld_issue(v4f& v, const float4* p) { asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory"); }
// up to 8 tagged units per thread: ALL loads in flight, one wait, only the stale ones asked again (one dependent hop instead of one per unit)
template <int N>
__device__ __forceinline__ bool poll_units(const float4* const (&p)[N], int n, float tag, float (&x)[N], float (&y)[N]) {
    static_assert(N == 8, "eight units per poll");
    v4f v[N];
    unsigned stale = (1u << n) - 1u, spins = 0;
    while (stale) {
#pragma unroll
        for (int k = 0; k < N; ++k) ld_issue(v[k], p[k]);      // (all eight every round: the registers are tied to the wait below)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
        for (int k = 0; k < N; ++k) if (((stale >> k) & 1u) && v[k].w == tag) { stale &= ~(1u << k); x[k] = v[k].x; y[k] = v[k].y; }
        if (stale) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 18)) return false; }
    }
    return true;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_probe(float4* partial /*[2][nb*spb]*/, float4* halo /*[2][nb*halo]*/, float* priv, float* result, int* timed_out, int iters, int spb, int halo_units, int priv_floats) {
    __shared__ float red[THREADS / 64];
    __shared__ int bad;
    const int b = blockIdx.x, nb = gridDim.x, V = nb * spb;
    float acc = 1.0f + b;
    float* mine = priv + (size_t)b * priv_floats;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int it = 1; it <= iters; ++it) {
        const float tag = (float)it;
        float4* P = partial + (size_t)(it & 1) * V;
        float4* H = halo + (size_t)(it & 1) * nb * halo_units;
        for (int k = threadIdx.x; k < priv_floats / 4; k += THREADS) { float4 v = reinterpret_cast<float4*>(mine)[k]; v.x = v.x * 0.999f + acc * 1e-6f; reinterpret_cast<float4*>(mine)[k] = v; acc += v.x * 1e-9f; }
        // publish: halo units and partials, each with its tag inside
        for (int k = threadIdx.x; k < halo_units; k += THREADS) st_sc1(H + (size_t)b * halo_units + k, make_float4(acc, acc, acc, tag));
        if ((int)threadIdx.x < spb) st_sc1(P + (size_t)b * spb + threadIdx.x, make_float4(acc * 1e-3f, 1.0f, 0.0f, tag));
        // consume: all partials (fixed order per thread), the two neighbours' halo units
        float sum = 0.0f;
        int good = 1;
        {
            const float4* pp[8]; float x[8], y[8]; int n = 0;
            for (int k = threadIdx.x; k < V && n < 8; k += THREADS) pp[n++] = P + k;
            for (int k = n; k < 8; ++k) pp[k] = P;
            if (n) { good = poll_units<8>(pp, n, tag, x, y); for (int k = 0; k < n; ++k) sum += y[k]; }
        }
        if (good) {
            const float4* pp[8]; float x[8], y[8]; int n = 0;
            for (int side = 0; side < 2; ++side) {
                const int nbk = (b + (side ? 1 : nb - 1)) % nb;
                for (int k = threadIdx.x; k < halo_units && n < 8; k += THREADS) pp[n++] = H + (size_t)nbk * halo_units + k;
            }
            for (int k = n; k < 8; ++k) pp[k] = P;
            if (n) { good = poll_units<8>(pp, n, tag, x, y); for (int k = 0; k < n; ++k) acc += x[k] * 1e-9f; }
        }
        if (!good) bad = 1;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (bad) { if (threadIdx.x == 0) *timed_out = 1; return; }
        float tot = 0.0f;
        for (int w = 0; w < THREADS / 64; ++w) tot += red[w];
        acc = acc * 0.5f + tot * 1e-6f;
        __syncthreads();
    }
    if (threadIdx.x == 0) result[b] = acc;
}

template <int THREADS>
static void run(int nb, int spb, int halo_units, int priv_kb, int iters) {
    float4 *partial, *halo; float *priv, *result; int* to;
    const int priv_floats = priv_kb * 256;
    hipMalloc(&partial, sizeof(float4) * 2 * nb * spb); hipMemset(partial, 0, sizeof(float4) * 2 * nb * spb);
    hipMalloc(&halo, sizeof(float4) * 2 * nb * (halo_units + 1)); hipMemset(halo, 0, sizeof(float4) * 2 * nb * (halo_units + 1));
    hipMalloc(&priv, sizeof(float) * (size_t)nb * (priv_floats + 4)); hipMemset(priv, 0, sizeof(float) * (size_t)nb * (priv_floats + 4));
    hipMalloc(&result, sizeof(float) * nb); hipMalloc(&to, 4); hipMemset(to, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(partial, 0, sizeof(float4) * 2 * nb * spb); hipMemset(halo, 0, sizeof(float4) * 2 * nb * (halo_units + 1));
        hipEventRecord(a);
        hipLaunchKernelGGL(k_probe<THREADS>, dim3(nb), dim3(THREADS), 0, 0, partial, halo, priv, result, to, iters, spb, halo_units, priv_floats);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    int h_to = 0; hipMemcpy(&h_to, to, 4, hipMemcpyDeviceToHost);
    printf("workgroups %4d x %4d threads, partials %4d (x %d per workgroup), halo %5d B, private %3d KB : %6.2f us per iteration%s\n", nb, THREADS, nb * spb, spb, halo_units * 16, priv_kb,
           best * 1e3f / iters, h_to ? "   TIMED OUT" : "");
    hipFree(partial); hipFree(halo); hipFree(priv); hipFree(result); hipFree(to);
}

int main() {
    const int iters = 400;
    // one resident workgroup per CU; 1 .. 4 virtual workgroups each (the headline solve has ~700 virtual workgroups)
    run<256>(256, 1, 0, 0, iters);
    run<256>(256, 1, 256, 0, iters);
    run<256>(256, 3, 256, 0, iters);
    run<256>(256, 4, 768, 0, iters);
    run<256>(256, 4, 768, 40, iters);
    run<1024>(256, 4, 768, 40, iters);
    run<256>(512, 2, 384, 20, iters);
    run<512>(256, 3, 1024, 40, iters);
    run<1024>(256, 3, 1024, 0, iters);
    return 0;
}
