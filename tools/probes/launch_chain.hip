// Probe: cost of a chain of dependent launches on one stream -- plain launches vs a captured hipGraph -- for a grid shaped like the
// brick-mapped PCG iteration kernel (~800 blocks x 256 threads) whose blocks (a) return at once, (b) do one dependent global load + store.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_noop(const int* flag) { if (*flag) return; }
__global__ __launch_bounds__(256) void k_touch(const int* flag, float* a, int n) {
    if (*flag) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = a[i] * 1.0001f + 1.0f;
}
__global__ __launch_bounds__(256) void k_lds(const int* flag, float* a, int n) {
    __shared__ float buf[3700];     // 14.8 KB like the staged PCG tile pair
    if (threadIdx.x < 8) buf[threadIdx.x] = 1.0f;
    if (*flag) return;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = a[i] * buf[threadIdx.x & 7] + buf[(threadIdx.x * 7) % 3700];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* flag; float* a; const int n = 800 * 256;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&a, n * 4)); CK(hipMemsetAsync(flag, 0, 4, s)); CK(hipMemsetAsync(a, 0, n * 4, s));
    int one = 1; int* flag1; CK(hipMalloc(&flag1, 4)); CK(hipMemcpyAsync(flag1, &one, 4, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    const int L = 66, reps = 200;
    for (int grid : {800, 400, 100, 8}) for (int variant = 0; variant < 5; ++variant) {   // 0: active touch, 1: returns at once (flag set), 2: k_noop with flag clear
        auto enqueue = [&]() { for (int i = 0; i < L; ++i) { if (variant == 0) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, flag, a, n); else if (variant == 1) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, flag1, a, n); else if (variant == 2) hipLaunchKernelGGL(k_noop, dim3(grid), dim3(256), 0, s, flag); else hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, s, variant == 3 ? flag1 : flag, a, n); } };
        for (int w = 0; w < 5; ++w) enqueue();
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        double host_us = 0;
        for (int r = 0; r < reps; ++r) { auto h0 = std::chrono::steady_clock::now(); enqueue(); host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count(); if ((r & 7) == 7) CK(hipStreamSynchronize(s)); }
        CK(hipStreamSynchronize(s));
        double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * L);
        host_us /= (reps * L);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); enqueue(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, s)); if ((r & 7) == 7) CK(hipStreamSynchronize(s)); }
        CK(hipStreamSynchronize(s));
        double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * L);
        printf("grid %4d x 256, %s: plain %.2f us/launch (host enqueue alone %.2f), hipGraph %.2f us/launch\n", grid, variant == 0 ? "load+store          " : (variant == 1 ? "returns at once     " : (variant == 2 ? "one scalar load only" : (variant == 3 ? "14.8 KB LDS, returns " : "14.8 KB LDS, active  "))), plain, host_us, graph);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
