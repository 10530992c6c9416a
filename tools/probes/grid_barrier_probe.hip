// Probe: what does a grid-wide barrier cost on this part WITHOUT the agent-scope fences (L2 write-back / invalidate) of grid_barrier()
// in blub_pcg.hip.h -- relaxed agent-scope atomics only, i.e. usable when the bulk data of a workgroup stays private to it (or to its XCD) and
// only flags / a few scalars cross XCDs.  One workgroup per CU (or `blocks`), `iters` barriers, bounded spins.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier_probe.hip -o build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_probe(unsigned* counter, int* timed_out, float* partial, float* result, int iters, int mode, float* scratch) {
    __shared__ int ok;
    __shared__ float red;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // a little local work: touch a private 4 KB slice (stays in this CU's / XCD's caches)
        float v = scratch[(size_t)blockIdx.x * 1024 + threadIdx.x * 4 + (it & 3)];
        acc += v * 1.0001f;
        scratch[(size_t)blockIdx.x * 1024 + threadIdx.x * 4 + ((it + 1) & 3)] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // what grid_barrier() does today
            // publish this block's partial with an agent-scope atomic store (write-through past the XCD's L2), then arrive
            __hip_atomic_store(partial + blockIdx.x, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = gridDim.x * (unsigned)(it + 1);
            int good = 1; unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 20) || __hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { good = 0; break; }
            }
            if (!good) __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ok = good;
        }
        __syncthreads();
        if (!ok) return;
        // every block re-reduces all partials (agent-scope loads: they bypass the non-coherent caches)
        float s = 0.f;
        for (int k = threadIdx.x; k < (int)gridDim.x; k += 256) s += __hip_atomic_load(partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (threadIdx.x == 0) red = 0.f;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) atomicAdd(&red, s);
        __syncthreads();
        acc = acc * 0.5f + red * 1e-9f;
    }
    if (threadIdx.x == 0) result[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    for (int blocks : {256, 512, 1024}) for (int mode : {0, 1}) {
        unsigned* counter; int* to; float *partial, *result, *scratch;
        hipMalloc(&counter, 4); hipMalloc(&to, 4); hipMalloc(&partial, blocks * 4); hipMalloc(&result, blocks * 4); hipMalloc(&scratch, (size_t)blocks * 4096);
        hipMemset(scratch, 0, (size_t)blocks * 4096);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(counter, 0, 4); hipMemset(to, 0, 4);
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, counter, to, partial, result, iters, mode, scratch);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        int h_to = 0; hipMemcpy(&h_to, to, 4, hipMemcpyDeviceToHost);
        printf("blocks %4d  %s : %.2f us per barrier + reduction (%d iterations)%s\n", blocks, mode ? "with agent-scope release/acquire fences" : "relaxed agent-scope atomics only   ", best * 1e3f / iters, iters, h_to ? "  TIMED OUT" : "");
        hipFree(counter); hipFree(to); hipFree(partial); hipFree(result); hipFree(scratch);
    }
    return 0;
}
