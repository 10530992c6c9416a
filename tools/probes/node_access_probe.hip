// Probe (round 6): what bounds a divergent walk over 32-byte nodes on MI355X -- lane accesses, instructions x lines, or lines?
//   A: every lane reads its OWN random node as two 16-byte loads (the P2G walk)          B: lane PAIRS read the two halves of one node (one load per lane)
//   C: every lane reads ONE 16-byte piece of its own random node (half the bytes of A)   D: as A, but the nodes of a wave are 64 CONSECUTIVE nodes (coalesced)
// Each thread follows a chain of `hops` dependent nodes (next index stored in the node), like the list walk.  Prints ns per node visit per CU-equivalent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
struct alignas(32) Node { float a[3]; unsigned next; float b[4]; };
template <int MODE>
__global__ __launch_bounds__(256) void k_walk(const Node* __restrict__ nodes, const unsigned* __restrict__ start, int hops, float* __restrict__ out) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    const float4* nd = reinterpret_cast<const float4*>(nodes);
    float acc = 0.f;
    if (MODE == 1) {      // lane pairs: list = t / 2, half = t & 1
        unsigned cur = start[t >> 1];
        for (int h = 0; h < hops; ++h) {
            const float4 v = nd[2 * (size_t)cur + (t & 1)];
            const float other = __shfl_xor(v.w, 1, 64);
            const unsigned nxt = (t & 1) ? __float_as_uint(other) : __float_as_uint(v.w);
            acc += v.x + v.y;
            cur = nxt;
        }
    } else {
        unsigned cur = start[t];
        for (int h = 0; h < hops; ++h) {
            const float4 p = nd[2 * (size_t)cur];
            float4 r = p;
            if (MODE != 2) r = nd[2 * (size_t)cur + 1];
            acc += p.x + r.y;
            cur = __float_as_uint(p.w);
        }
    }
    out[t] = acc;
}
int main(int argc, char** argv) {
    const size_t N = (size_t)1 << 25;      // 32 M nodes = 1 GiB
    const int hops = 8;
    std::vector<Node> h(N);
    std::vector<unsigned> perm(N);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    // chains: node perm[i] -> perm[i + T] (T threads), so that hop k of thread t is perm[t + k T]: random per lane
    const size_t T = N / hops;
    for (size_t i = 0; i < N; ++i) { h[perm[i]].next = i + T < N ? perm[i + T] : perm[i % T]; h[perm[i]].a[0] = 1.f; h[perm[i]].b[1] = 1.f; }
    std::vector<unsigned> start_rand(perm.begin(), perm.begin() + T);
    // coalesced variant: chains i -> i + T in index order
    std::vector<Node> hc(N);
    for (size_t i = 0; i < N; ++i) { hc[i].next = i + T < N ? (unsigned)(i + T) : (unsigned)(i % T); }
    std::vector<unsigned> start_seq(T); std::iota(start_seq.begin(), start_seq.end(), 0u);
    Node *d, *dc; unsigned *s, *sc; float* out;
    hipMalloc(&d, N * sizeof(Node)); hipMalloc(&dc, N * sizeof(Node)); hipMalloc(&s, T * 4); hipMalloc(&sc, T * 4); hipMalloc(&out, T * 2 * 4);
    hipMemcpy(d, h.data(), N * sizeof(Node), hipMemcpyHostToDevice); hipMemcpy(dc, hc.data(), N * sizeof(Node), hipMemcpyHostToDevice);
    hipMemcpy(s, start_rand.data(), T * 4, hipMemcpyHostToDevice); hipMemcpy(sc, start_seq.data(), T * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int mode, const Node* nd, const unsigned* st, size_t threads) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_walk<0>, dim3(threads / 256), dim3(256), 0, 0, nd, st, hops, out);
            if (mode == 1) hipLaunchKernelGGL(k_walk<1>, dim3(threads / 256), dim3(256), 0, 0, nd, st, hops, out);
            if (mode == 2) hipLaunchKernelGGL(k_walk<2>, dim3(threads / 256), dim3(256), 0, 0, nd, st, hops, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        const double visits = (double)T * hops;
        printf("%-44s %8.3f ms  %6.2f ps per node visit  (%5.2f CU-cycles per visit per CU at 2.4 GHz x 256 CUs)\n", name, best, best * 1e9 / visits, best * 1e-3 * 2.4e9 * 256 / visits);
    };
    run("A own node, 2 x 16 B per lane (random)", 0, d, s, T);
    run("B lane pairs, 1 x 16 B per lane (random)", 1, d, s, 2 * T);
    run("C own node, 1 x 16 B per lane (random)", 2, d, s, T);
    run("D own node, 2 x 16 B per lane (consecutive)", 0, dc, sc, T);
    run("E lane pairs (consecutive)", 1, dc, sc, 2 * T);
    return 0;
}
