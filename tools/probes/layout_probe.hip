// Probe: how much faster would the brick-mapped PCG iteration move its data if the solver fields were stored brick-major (2 KB per
// brick and field, contiguous) instead of as rows of the linear 256^3 volume (32 rows of 64 B per brick and field, 1 KB apart)?
// One launch = 1400 bricks, 5 fields read + 5 written per own quad (the single-reduction kernel's own traffic), chained launches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NX = 256, NY = 256, NZ = 256, BX = 16, BY = 8, BZ = 4;
struct Ptrs { float* f[10]; };
template <bool PACKED, int HALO>
__global__ __launch_bounds__(256) void k_touch(const uint32_t* __restrict__ list, int n, Ptrs P) {
    const int half = threadIdx.x >> 7, t = threadIdx.x & 127;
    const int i = blockIdx.x * 2 + half;
    if (i >= n) return;
    const uint32_t b = list[i];
    const int nbx = NX / BX, nby = NY / BY;
    const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
    float4 acc = make_float4(0, 0, 0, 0);
    size_t own;
    if (PACKED) own = ((size_t)i * 128 + t) * 4;
    else own = ((size_t)(bz * BZ + (t >> 5)) * NY + (by * BY + ((t >> 2) & 7))) * NX + bx * BX + ((t & 3) << 2);
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float4 v = *reinterpret_cast<const float4*>(P.f[k] + own); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (HALO) {   // the y / z face-halo rows of three fields (112 quads per brick: threads 0..111), always from the linear / neighbour location
        if (t < 112) {
            const int row = t >> 2, q = t & 3;          // 28 halo rows: 2 x 4 (y faces) + 2 x 8 ... approximated: rows around the brick
            int yy, zz;
            if (row < 8) { yy = (row & 1) ? BY : -1; zz = row >> 1; } else { const int r2 = row - 8; yy = r2 % 8; zz = (r2 / 8) & 1 ? BZ : -1; if (r2 >= 16) { yy = (r2 - 16) % 8; zz = -1; } }
            int gy = by * BY + yy, gz = bz * BZ + zz;
            gy = min(max(gy, 0), NY - 1); gz = min(max(gz, 0), NZ - 1);
            size_t h;
            if (PACKED) {   // neighbour brick's packed chunk: emulate with a different brick's slot (same access shape: 64 B rows inside a 2 KB chunk)
                const int j = (i + 1 + (row & 3)) % n;
                h = ((size_t)j * 128 + ((row * 4 + q) & 127)) * 4;
            } else h = ((size_t)gz * NY + gy) * NX + bx * BX + (q << 2);
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float4 v = *reinterpret_cast<const float4*>(P.f[k] + h); acc.x += v.x; acc.y += v.y; }
        }
    }
#pragma unroll
    for (int k = 5; k < 10; ++k) *reinterpret_cast<float4*>(P.f[k] + own) = acc;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t N = (size_t)NX * NY * NZ;
    Ptrs P;
    for (int k = 0; k < 10; ++k) { CK(hipMalloc(&P.f[k], N * 4)); CK(hipMemsetAsync(P.f[k], 0, N * 4, s)); }
    // a compact blob of ~1400 bricks in two corners like the headline scene
    std::vector<uint32_t> bricks;
    const int nbx = NX / BX, nby = NY / BY, nbz = NZ / BZ;
    for (int bz = 0; bz < nbz; ++bz) for (int by = 0; by < nby; ++by) for (int bx = 0; bx < nbx; ++bx) {
        const bool a = bx < 5 && by < 7 && bz < 20, c = bx >= nbx - 5 && by < 7 && bz >= nbz - 20;
        if (a || c) bricks.push_back((uint32_t)((bz * nby + by) * nbx + bx));
    }
    const int n = (int)bricks.size();
    uint32_t* list; CK(hipMalloc(&list, n * 4)); CK(hipMemcpyAsync(list, bricks.data(), n * 4, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    const int L = 66, reps = 60;
    auto run = [&](auto kernel, const char* name) {
        const dim3 grid((n + 1) / 2), block(256);
        for (int w = 0; w < 3; ++w) for (int i = 0; i < L; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, s, (const uint32_t*)list, n, P);
        (void)hipStreamSynchronize(s);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < L; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, s, (const uint32_t*)list, n, P);
        (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * L);
        printf("%-44s %d bricks: %.2f us per launch\n", name, n, us);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    };
    run(k_touch<false, 0>, "linear rows, own quads only");
    run(k_touch<true, 0>, "brick-major, own quads only");
    run(k_touch<false, 1>, "linear rows, own + y/z halo rows");
    run(k_touch<true, 1>, "brick-major, own + halo rows");
    return 0;
}
