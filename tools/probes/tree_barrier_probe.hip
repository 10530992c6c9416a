// Probe 2: what would ONE iteration of a persistent brick-mapped PCG cost in synchronisation + exchange, with the recipe that avoids the agent-scope
// fences (cdna_hip_programming.md G16): exchanged data through write-through (sc1 / agent-scope) stores and loads, bulk data private to the workgroup.
//   * two-level barrier: workgroups arrive at one of `groups` counters, the last arriver of a group arrives at the root, the last arriver at the root
//     publishes the epoch to every group's release word; workgroups poll their group's word (relaxed agent-scope loads, bounded);
//   * every workgroup publishes a 16-byte partial before arriving and re-reduces ALL partials after the release (as the launched kernels do);
//   * `halo` floats per workgroup are written write-through before the barrier and read from the NEXT workgroup's slice after it;
//   * private work: a 40 KB slice per workgroup read and rewritten with plain accesses.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tree_barrier_probe.hip -o build/tree_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Sync { unsigned group_count[64]; unsigned root_count; unsigned pad[15]; unsigned release[64 * 16]; int timed_out; };

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

__global__ __launch_bounds__(256) void k_probe(Sync* S, float4* partial, float4* halo, float* priv, float* result, int iters, int groups, int halo_quads, int priv_floats) {
    __shared__ int ok;
    __shared__ float red[4];
    const int b = blockIdx.x, nb = gridDim.x, g = b % groups;
    const int gsize = nb / groups + (g < nb % groups ? 1 : 0);
    float acc = 1.0f + b;
    float* mine = priv + (size_t)b * priv_floats;
    for (int it = 0; it < iters; ++it) {
        // private bulk work (plain accesses)
        for (int k = threadIdx.x; k < priv_floats / 4; k += 256) { float4 v = reinterpret_cast<float4*>(mine)[k]; v.x = v.x * 0.999f + acc * 1e-6f; reinterpret_cast<float4*>(mine)[k] = v; acc += v.x * 1e-9f; }
        // publish halo + partial write-through
        for (int k = threadIdx.x; k < halo_quads; k += 256) st_sc1(halo + (size_t)b * halo_quads + k, make_float4(acc, acc, acc, (float)it));
        if (threadIdx.x == 0) st_sc1(partial + b, make_float4(acc, 0.f, 0.f, (float)it));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned epoch = (unsigned)(it + 1);
            int good = 1;
            const unsigned a = __hip_atomic_fetch_add(&S->group_count[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a + 1u == (unsigned)gsize * epoch) {
                const unsigned r = __hip_atomic_fetch_add(&S->root_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r + 1u == (unsigned)groups * epoch)
                    for (int k = 0; k < groups; ++k) __hip_atomic_store(&S->release[k * 16], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned spins = 0;
            while (__hip_atomic_load(&S->release[g * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 18)) { good = 0; break; }
            }
            if (!good) __hip_atomic_store(&S->timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = good;
        }
        __syncthreads();
        if (!ok) return;
        // re-reduce all partials + read the neighbour's halo (write-through data: sc1 loads)
        float s = 0.f;
        {   // up to 6 partial loads + 1 halo load per thread in flight, ONE wait
            v4f v[7];
            const int nbr = (b + 1) % nb;
#pragma unroll
            for (int j = 0; j < 6; ++j) { const int k = min((int)threadIdx.x + 256 * j, nb - 1); asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v[j]) : "v"(partial + k) : "memory"); }
            { const int k = min((int)threadIdx.x, max(halo_quads - 1, 0)); asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v[6]) : "v"(halo + (size_t)nbr * halo_quads + k) : "memory"); }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]) :: "memory");
#pragma unroll
            for (int j = 0; j < 6; ++j) if ((int)threadIdx.x + 256 * j < nb) s += v[j].x;
            if ((int)threadIdx.x < halo_quads) s += v[6].x * 1e-9f;
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        acc = acc * 0.5f + (red[0] + red[1] + red[2] + red[3]) * 1e-9f;
        __syncthreads();
    }
    if (threadIdx.x == 0) result[b] = acc;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    struct Cfg { int blocks, groups, halo_quads, priv_floats; };
    const Cfg cfgs[] = {{256, 8, 256, 10240}, {256, 16, 256, 10240}, {650, 8, 256, 10240}, {650, 26, 256, 10240}, {650, 26, 0, 0}, {650, 26, 256, 0}, {650, 64, 256, 10240}, {1024, 32, 256, 10240}, {1300, 36, 256, 5120}};
    for (const Cfg& c : cfgs) {
        Sync* S; float4 *partial, *halo; float *priv, *result;
        hipMalloc(&S, sizeof(Sync)); hipMalloc(&partial, c.blocks * 16); hipMalloc(&halo, (size_t)c.blocks * (c.halo_quads + 1) * 16);
        hipMalloc(&priv, (size_t)c.blocks * (c.priv_floats + 4) * 4); hipMalloc(&result, c.blocks * 4);
        hipMemset(priv, 0, (size_t)c.blocks * (c.priv_floats + 4) * 4); hipMemset(halo, 0, (size_t)c.blocks * (c.halo_quads + 1) * 16); hipMemset(partial, 0, c.blocks * 16);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(S, 0, sizeof(Sync));
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(k_probe, dim3(c.blocks), dim3(256), 0, 0, S, partial, halo, priv, result, iters, c.groups, c.halo_quads, c.priv_floats);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        Sync h; hipMemcpy(&h, S, sizeof(Sync), hipMemcpyDeviceToHost);
        printf("blocks %4d groups %2d halo %4d B private %3d KB : %.2f us per iteration (%d iterations)%s\n", c.blocks, c.groups, c.halo_quads * 16, c.priv_floats * 4 / 1024, best * 1e3f / iters, iters, h.timed_out ? "  TIMED OUT" : "");
        hipFree(S); hipFree(partial); hipFree(halo); hipFree(priv); hipFree(result);
    }
    return 0;
}
