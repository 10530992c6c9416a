// Probe (round 6): what does it cost to write HALF of every 32-byte gather node (16 bytes at a 32-byte stride) instead of a contiguous float4 array?
// (k_advect could write the row halves of the nodes directly and k_build_lists only the {position, link} halves: no 96-byte copy per particle.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
__global__ void k_contig(float4* a, float4* b, float4* c, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) { float4 v = make_float4(i, 1, 2, 3); a[i] = v; b[i] = v; c[i] = v; } }
__global__ void k_half(float4* a, float4* b, float4* c, size_t n, int half) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) { float4 v = make_float4(i, 1, 2, 3); a[2 * i + half] = v; b[2 * i + half] = v; c[2 * i + half] = v; } }
__global__ void k_read_nodes(const float4* a, size_t n, float* out) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) { float4 p = a[2 * i], r = a[2 * i + 1]; if (p.x + r.y == -1.f) out[0] = 1.f; } }
int main() {
    const size_t n = (size_t)1 << 25;      // 32 M particles
    float4 *a, *b, *c; float* out;
    hipMalloc(&a, n * 32); hipMalloc(&b, n * 32); hipMalloc(&c, n * 32); hipMalloc(&out, 4);
    hipMemset(a, 0, n * 32); hipMemset(b, 0, n * 32); hipMemset(c, 0, n * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t = [&](const char* name, auto launch, double useful) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
        printf("%-46s %7.3f ms  %6.0f GB/s of useful bytes\n", name, best, useful / best * 1e-6);
    };
    const dim3 g((unsigned)(n / 256));
    t("3 contiguous float4 arrays (48 B / particle)", [&] { hipLaunchKernelGGL(k_contig, g, dim3(256), 0, 0, a, b, c, n); }, 48.0 * n);
    t("row halves of 3 node arrays (16 B at stride 32)", [&] { hipLaunchKernelGGL(k_half, g, dim3(256), 0, 0, a, b, c, n, 1); }, 48.0 * n);
    t("position halves of 3 node arrays", [&] { hipLaunchKernelGGL(k_half, g, dim3(256), 0, 0, a, b, c, n, 0); }, 48.0 * n);
    t("read both halves of one node array (32 B)", [&] { hipLaunchKernelGGL(k_read_nodes, g, dim3(256), 0, 0, a, n, out); }, 32.0 * n);
    return 0;
}
