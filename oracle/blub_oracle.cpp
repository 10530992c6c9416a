// =====================================================================================================
// blub_oracle.cpp -- CPU restatement of Wumpf/blub's HybridFluid::step hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg may load this library.  The product (blub_amd/, libblubhip.so) never links,
// imports or calls anything in oracle/.
//
// PARITY PINNED (round 4) AGAINST THE REFERENCE'S OWN SHADER TEXT: the reference ships no tests or golden
// vectors and its wgpu/Vulkan path cannot run in this image (no Rust, no shaderc, no Vulkan ICD), but its
// arithmetic lives in GLSL that g++ can compile: oracle/glsl/ (glsl_shim.h + glsl2cpp.py, a lexical
// preprocessor) builds the UNMODIFIED shader/simulation/**/*.comp into oracle/_ref/libblubref.so and
// oracle/glsl/ref_fluid.py dispatches them as HybridFluid::step records them.  With `dot_mode 2` (the
// reference's own reduction order) this file reproduces that library BIT FOR BIT -- every stage of a step
// incl. moving solids, both PCG solves with both readings of Q1, the literal Q4 binning
// (tests/test_oracle_vs_ref.py: committed fixtures tests/golden/ref_*.npz + live runs where
// /root/reference exists).  What remains implementation-defined in Vulkan and is therefore a CHOICE here:
// the hardware trilinear filter (evaluated separably in f32; bounded against the two alternatives), the
// order of atomics (ascending invocation index), out-of-range LOD (Q1 switch), f32 division / sqrt rounding.
//
// This file is a literal restatement of the reference's compute shaders (every function cites the
// file:line it follows, relative to /root/reference) with the GPU image semantics of SURVEY.md Appendix A:
//   * out-of-bounds image/texel reads return 0, OOB stores are dropped
//   * marker R8Snorm {+1 FLUID, -1 AIR, 0 SOLID} held as int8
//   * all buffers/volumes zero-initialised
//   * linked-list insertion order = ascending particle index (the GPU order is a race)
//   * plain IEEE f32 arithmetic, no FMA contraction (build with -ffp-contract=off); dot-product
//     reductions: f64 per z-plane by default (deterministic for any thread count), or the reference's tree
// Also checked against: analytic known-answer tests (tests/test_oracle_kat.py), scipy's sparse solver for
// the PCG, the published xoshiro256++/SplitMix64 vectors for the seeding RNG (the one part still unpinned:
// rand 0.8.5 is not on disk -- parity tests never depend on it).
// =====================================================================================================
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int8_t CELL_SOLID = 0;   // shader/simulation/hybrid_fluid.glsl:20
constexpr int8_t CELL_FLUID = 1;   // :21
constexpr int8_t CELL_AIR = -1;    // :22
constexpr uint32_t INVALID_LL = 0xFFFFFFFFu;  // shader/simulation/particles.glsl:3

struct F4 { float x, y, z, w; };
struct PosLl { float x, y, z; uint32_t ll; };   // ParticlePositionLl, hybrid_fluid.rs:76-85

enum Volume { V_MARKER = 0, V_LL = 1, V_VELX = 2, V_VELY = 3, V_VELZ = 4, V_PRESSURE_VELOCITY = 5, V_PRESSURE_DENSITY = 6,
              V_RESIDUAL = 7, V_SEARCH = 8, V_AUX = 9, V_AUX_TEMP = 10, V_SOLID = 11 };
enum Stage { ST_TRANSFER = 0, ST_DIVERGENCE = 1, ST_SOLVE_VELOCITY = 2, ST_BINNING = 3, ST_PROJECT = 4, ST_ADVECT = 5,
             ST_DENSITY_GATHER = 6, ST_SOLVE_DENSITY = 7, ST_POSITION_CHANGE = 8, ST_CORRECT = 9 };
enum PrecondMode { PRECOND_ZERO = 0, PRECOND_LOD0 = 1 };
enum BinningMode { BINNING_FIXED = 0, BINNING_LITERAL = 1, BINNING_OFF = 2 };   // Q4

struct SolverConfig { float error_tolerance = 0.1f; int max_num_iterations = 32; int error_check_frequency = 4; };  // hybrid_fluid.rs:253-257
struct SolverStats { float error = 0; int iterations = 0; };

inline float satf(float v) { return std::min(std::max(v, 0.0f), 1.0f); }
inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }   // GLSL mix
inline float fractf(float v) { return v - std::floor(v); }                         // GLSL fract
inline float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

struct Oracle {
    int nx, ny, nz;
    size_t N;
    uint32_t max_particles, num_particles = 0;
    float gravity[3] = {0, 0, 0};
    std::vector<PosLl> pos, pos_tmp;
    std::vector<F4> pvel[3];
    std::vector<int8_t> marker;
    std::vector<uint32_t> ll;
    std::vector<float> vel[3], pressure[2], residual, search, aux, aux_temp;
    std::vector<F4> solid;  // empty => all-zero voxelisation (scene/voxelization.rs:125)
    SolverConfig cfg[2];
    SolverStats last_stats[2];
    bool pressure_cleared[2] = {false, false};
    int dot_mode = 0;   // 2: the reference's own reduction order, literally (reduce_literal below; bit-exact against oracle/_ref);
                        // 0: dot products accumulated in f64 (default); 1: in f32 (row sums -> plane sums -> total): a sensitivity probe for
                        // the reference's own f32 tree reductions (pressure_reduce.comp:37-61), see tests/test_oracle_kat.py
    int precond_mode = PRECOND_ZERO;
    int binning_mode = BINNING_FIXED;
    uint32_t rebin_freq = 60;         // hybrid_fluid.rs:603-605
    uint32_t step_counter = 0;
    uint64_t solver_iterations_total = 0;
    double solver_seconds_total = 0;

    Oracle(int x, int y, int z, uint32_t maxp) : nx(x), ny(y), nz(z), N((size_t)x * y * z), max_particles(maxp) {
        pos.assign(maxp, PosLl{0, 0, 0, 0});
        pos_tmp.assign(maxp, PosLl{0, 0, 0, 0});
        for (auto& v : pvel) v.assign(maxp, F4{0, 0, 0, 0});
        marker.assign(N, 0);
        ll.assign(N, 0);
        for (auto& v : vel) v.assign(N, 0.f);
        for (auto& v : pressure) v.assign(N, 0.f);
        residual.assign(N, 0.f); search.assign(N, 0.f); aux.assign(N, 0.f); aux_temp.assign(N, 0.f);
    }

    inline bool inb(int x, int y, int z) const { return (unsigned)x < (unsigned)nx && (unsigned)y < (unsigned)ny && (unsigned)z < (unsigned)nz; }
    inline size_t idx(int x, int y, int z) const { return ((size_t)z * ny + y) * nx + x; }
    inline int8_t mk(int x, int y, int z) const { return inb(x, y, z) ? marker[idx(x, y, z)] : CELL_SOLID; }
    inline float fv(const std::vector<float>& v, int x, int y, int z) const { return inb(x, y, z) ? v[idx(x, y, z)] : 0.f; }
    inline F4 sv(int x, int y, int z) const { return (solid.empty() || !inb(x, y, z)) ? F4{0, 0, 0, 0} : solid[idx(x, y, z)]; }
    inline float sv_c(int x, int y, int z, int c) const { F4 s = sv(x, y, z); return c == 0 ? s.x : (c == 1 ? s.y : s.z); }

    // ---- seeding: hybrid_fluid.rs:609-678 -------------------------------------------------------
    // rand 0.8.5 SmallRng (64-bit) = xoshiro256++; SmallRng does not forward seed_from_u64, so the
    // rand_core 0.6 default applies: a PCG32 stream fills the 32-byte seed (SURVEY 8c "Seeding RNG").
    struct SmallRng {
        uint64_t s[4];
        static inline uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
        void from_seed_bytes(const uint8_t* b) {
            for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int k = 0; k < 8; ++k) v |= (uint64_t)b[i * 8 + k] << (8 * k); s[i] = v; }
            if (!(s[0] | s[1] | s[2] | s[3])) seed_from_u64(0);
        }
        void seed_from_u64(uint64_t state) {
            const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
            uint8_t seed[32];
            for (int c = 0; c < 8; ++c) {
                state = state * MUL + INC;
                uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
                uint32_t rot = (uint32_t)(state >> 59);
                uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
                for (int k = 0; k < 4; ++k) seed[c * 4 + k] = (uint8_t)(x >> (8 * k));
            }
            from_seed_bytes(seed);
        }
        uint64_t next_u64() {
            uint64_t result = rotl(s[0] + s[3], 23) + s[0];
            uint64_t t = s[1] << 17;
            s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
            return result;
        }
        uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
        float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }   // Standard f32: 24 bits
    };

    static uint32_t f32_as_u32_sat(float v) {  // Rust `as u32`: saturating, NaN -> 0
        if (!(v > 0.0f)) return 0;
        if (v >= 4294967296.0f) return 0xFFFFFFFFu;
        return (uint32_t)v;
    }
    void clamp_to_grid(const float* g, uint32_t* out) const {   // hybrid_fluid.rs:609-617
        const uint32_t dim[3] = {(uint32_t)nx, (uint32_t)ny, (uint32_t)nz};
        for (int k = 0; k < 3; ++k) out[k] = std::max(std::min(dim[k] - 1, f32_as_u32_sat(g[k])), 1u);
    }
    int add_fluid_cube(const float* min_grid, const float* max_grid) {   // hybrid_fluid.rs:620-678
        uint32_t mn[3], mx[3], ext[3];
        clamp_to_grid(min_grid, mn); clamp_to_grid(max_grid, mx);
        for (int k = 0; k < 3; ++k) ext[k] = mx[k] - mn[k];
        uint32_t num_new = ext[0] * ext[1] * ext[2] * 8u;
        if (max_particles < num_new + num_particles) num_new = max_particles - num_particles;   // :627-633 truncation
        SmallRng rng; rng.seed_from_u64((uint64_t)(num_particles + num_new));                    // :637
        for (uint32_t i = 0; i < num_new; ++i) {
            float cx = (float)(mn[0] + i / 8 % ext[0]);
            float cy = (float)(mn[1] + i / 8 / ext[0] % ext[1]);
            float cz = (float)(mn[2] + i / 8 / ext[0] / ext[1]);
            uint32_t sidx = i % 8;
            float rx = rng.gen_f32(), ry = rng.gen_f32(), rz = rng.gen_f32();   // cgmath Vector3: x, y, z
            float ox = (float)(sidx % 2) * 0.5f + rx * 0.5f;
            float oy = (float)(sidx / 2 % 2) * 0.5f + ry * 0.5f;
            float oz = (float)(sidx / 4 % 2) * 0.5f + rz * 0.5f;
            pos[num_particles + i] = PosLl{cx + ox, cy + oy, cz + oz, INVALID_LL};
        }
        num_particles += num_new;
        return (int)num_new;
    }

    // ---- T1: transfer_clear.comp:10-14 ----------------------------------------------------------
    void transfer_clear(int comp) {
        std::fill(ll.begin(), ll.end(), 0u);
        if (comp == 0) std::fill(marker.begin(), marker.end(), CELL_AIR);
    }
    // ---- T2: transfer_build_linkedlist.comp:10-26 ------------------------------------------------
    void build_linkedlist(int comp) {
        for (uint32_t i = 0; i < num_particles; ++i) {
            PosLl& p = pos[i];
            if (comp == 0) { int x = (int)p.x, y = (int)p.y, z = (int)p.z; if (inb(x, y, z)) marker[idx(x, y, z)] = CELL_FLUID; }
            float off[3] = {0.5f, 0.5f, 0.5f}; off[comp] = 1.0f;
            int dx = (int)(p.x - off[0]), dy = (int)(p.y - off[1]), dz = (int)(p.z - off[2]);
            uint32_t old = 0;
            if (inb(dx, dy, dz)) { size_t c = idx(dx, dy, dz); old = ll[c]; ll[c] = i + 1; }
            p.ll = old - 1u;
        }
    }
    // ---- T3: transfer_set_boundary_marker.comp:11-19 ---------------------------------------------
    void set_boundary_marker() {
#pragma omp parallel for
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            if (x == 0 || y == 0 || z == 0 || x == nx - 1 || y == ny - 1 || z == nz - 1) marker[idx(x, y, z)] = CELL_SOLID;
            else if (!solid.empty() && solid[idx(x, y, z)].w != 0.0f) marker[idx(x, y, z)] = CELL_SOLID;
        }
    }
    // ---- T4: transfer_gather_velocity.comp:39-127 -------------------------------------------------
    // Per non-border thread (one per cell g): 8 lists = heads at g - {0,1}^3, consumed round-major, <=12 rounds (:61).
    void gather_velocity(int comp, float dt) {
        static const int OFF[8][3] = {{0,0,0},{1,0,0},{0,1,0},{1,1,0},{0,0,1},{1,0,1},{0,1,1},{1,1,1}};   // :57-93 order
        std::vector<float>& out = vel[comp];
        const std::vector<F4>& rows = pvel[comp];
#pragma omp parallel for schedule(dynamic, 1)
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            int nb[3] = {x, y, z}; nb[comp] += 1;
            int8_t mA = mk(x, y, z), mB = mk(nb[0], nb[1], nb[2]);
            bool writes = (mA == CELL_FLUID || mB == CELL_FLUID);            // :50
            bool computes = (mA != CELL_SOLID && mB != CELL_SOLID);          // :51
            if (!writes) continue;
            float v = 0.f, wsum = 0.f;
            if (computes) {
                float sp[3] = {(float)x + 0.5f, (float)y + 0.5f, (float)z + 0.5f}; sp[comp] += 0.5f;   // :53-54
                uint32_t cur[8];
                for (int k = 0; k < 8; ++k) { int gx = x - OFF[k][0], gy = y - OFF[k][1], gz = z - OFF[k][2];
                    cur[k] = (inb(gx, gy, gz) ? ll[idx(gx, gy, gz)] : 0u) - 1u; }
                for (int round = 0; round < 12; ++round) {
                    bool any = false;
                    for (int k = 0; k < 8; ++k) {
                        uint32_t pi = cur[k];
                        if (pi == INVALID_LL) continue;
                        any = true;
                        const PosLl& p = pos[pi]; const F4& r = rows[pi];
                        cur[k] = p.ll;
                        float tx = sp[0] - p.x, ty = sp[1] - p.y, tz = sp[2] - p.z;                     // :20
                        float ox = satf(1.0f - std::fabs(tx)), oy = satf(1.0f - std::fabs(ty)), oz = satf(1.0f - std::fabs(tz));
                        float w = ox * oy * oz;                                                         // :22
                        float d = ((r.x * tx + r.y * ty) + r.z * tz) + r.w * 1.0f;                       // :24 dot(row, vec4(d,1))
                        v += w * d; wsum += w;
                    }
                    if (!any) break;
                }
                if (wsum > 0.0f) v /= wsum;                                   // :118-119
                v += gravity[comp] * dt;                                      // :120
            } else v = 0.0f;                                                  // :121-124
            out[idx(x, y, z)] = v;
        }
    }
    // ---- D1: divergence_compute.comp:28-87 --------------------------------------------------------
    void divergence_compute() {
#pragma omp parallel for
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            if (mk(x, y, z) != CELL_FLUID) continue;
            float px = fv(vel[0], x, y, z), py = fv(vel[1], x, y, z), pz = fv(vel[2], x, y, z);
            float qx = fv(vel[0], x - 1, y, z), qy = fv(vel[1], x, y - 1, z), qz = fv(vel[2], x, y, z - 1);
            float div = px - qx; div += py - qy; div += pz - qz;             // :60-62
            auto wall = [&](int gx, int gy, int gz, float wallv, int c) -> float {   // :20-26
                return mk(gx, gy, gz) == CELL_SOLID ? wallv - sv_c(gx, gy, gz, c) : 0.0f; };
            div += wall(x - 1, y, z, qx, 0); div += wall(x, y - 1, z, qy, 1); div += wall(x, y, z - 1, qz, 2);   // :67-75
            div -= wall(x + 1, y, z, px, 0); div -= wall(x, y + 1, z, py, 1); div -= wall(x, y, z + 1, pz, 2);   // :76-84
            residual[idx(x, y, z)] = div;
        }
    }

    // ---- PCG: pressure_solver.rs:591-729 + shader/simulation/pressure_solver/*  ----
    struct Nb { int8_t m[6]; };
    inline Nb nbm(int x, int y, int z) const { return Nb{{mk(x - 1, y, z), mk(x + 1, y, z), mk(x, y - 1, z), mk(x, y + 1, z), mk(x, y, z - 1), mk(x, y, z + 1)}}; }
    // MultiplyWithCoefficientMatrix, pressure.glsl:34-75
    inline float mulA(const std::vector<float>& t, int x, int y, int z, float c) const {
        Nb n = nbm(x, y, z);
        float d = 0.f; for (int k = 0; k < 6; ++k) d += std::fabs((float)n.m[k]);
        float r = 0.f; r += d * c;
        if (n.m[0] == CELL_FLUID) r -= t[idx(x - 1, y, z)];
        if (n.m[1] == CELL_FLUID) r -= t[idx(x + 1, y, z)];
        if (n.m[2] == CELL_FLUID) r -= t[idx(x, y - 1, z)];
        if (n.m[3] == CELL_FLUID) r -= t[idx(x, y + 1, z)];
        if (n.m[4] == CELL_FLUID) r -= t[idx(x, y, z - 1)];
        if (n.m[5] == CELL_FLUID) r -= t[idx(x, y, z + 1)];
        return r;
    }
    // pressure_apply_preconditioner.comp:36-82; returns sum(out*r) when with_dot
    double precond_pass(const std::vector<float>& in, std::vector<float>& out, bool with_dot) {
        std::vector<double> part(nz, 0.0);
        const bool f32dots = dot_mode == 1;
        const bool literal = dot_mode == 2 && with_dot;
        if (literal) redbuf.assign(N, 0.0f);
#pragma omp parallel for
        for (int z = 0; z < nz; ++z) { double acc = 0; float accp = 0.f; for (int y = 0; y < ny; ++y) { float accr = 0.f; for (int x = 0; x < nx; ++x) {
            if (mk(x, y, z) != CELL_FLUID) continue;
            size_t c = idx(x, y, z);
            float res = in[c];
            Nb n = nbm(x, y, z);
            if (precond_mode == PRECOND_LOD0) {     // neighbour fetches at lod 1 (:58,61,64): Q1 -- "zero" reading drops them
                if (n.m[0] == CELL_FLUID) res -= in[idx(x - 1, y, z)];
                if (n.m[2] == CELL_FLUID) res -= in[idx(x, y - 1, z)];
                if (n.m[4] == CELL_FLUID) res -= in[idx(x, y, z - 1)];
            }
            float d = 0.f; for (int k = 0; k < 6; ++k) d += (n.m[k] != CELL_SOLID) ? 1.0f : 0.0f;
            if (d > 0.0f) res /= d;
            out[c] = res;
            if (literal) { size_t a = reduce_addr(x, y, z); if (a < N) redbuf[a] = res * residual[c]; }
            else if (with_dot) { if (f32dots) accr += res * residual[c]; else acc += (double)(res * residual[c]); }
        } accp += accr; } part[z] = f32dots ? (double)accp : acc; }
        if (literal) return (double)reduce_literal(false);
        if (f32dots) { float s = 0.f; for (double p : part) s += (float)p; return (double)s; }
        double s = 0; for (double p : part) s += p; return s;
    }
    // ---- dot_mode 2: the reference's own reduction, operation for operation ------------------------
    // Level 0: every invocation of an 8x8x1-workgroup grid kernel writes its term to the WG-linear address
    // GetReduceBufferAddress() (pressure_apply_coeff.comp:13-17); reduce_addr() is that address for cell (x, y, z).
    // Levels 1..: pressure_reduce.comp:35-61 -- thread t of workgroup w starts from 0.0 and folds in the <= 16 elements
    // gid + k * (1024 * numWG), then a 1024 -> 2 LDS tree (stride 512 .. 2) and shared[0] (op) shared[1]; dispatch sizes and
    // SourceBufferSize as pressure_solver.rs:543-589 / pressure_init.comp:27-31 compute them (Q5: every indirect level uses
    // DispatchCommandReduce0).
    inline size_t reduce_addr(int x, int y, int z) const {
        const int gwx = (nx + 7) / 8, gwy = (ny + 7) / 8;
        return (size_t)((x & 7) + 8 * (y & 7)) + 64u * (((size_t)z * gwy + (size_t)(y >> 3)) * gwx + (size_t)(x >> 3));
    }
    std::vector<float> redbuf;   // level-0 buffer ("Buffer: DotProduct Reduce 0", N floats: addresses >= N are dropped, robust access)
    static float reduce_workgroup(const std::vector<float>& src, size_t size, uint32_t wg, uint32_t num_wg, bool is_max) {
        float sh[1024];
        for (uint32_t t = 0; t < 1024; ++t) {
            size_t addr = (size_t)wg * 1024 + t;
            float v = 0.0f;
            for (int k = 0; k < 16; ++k) { if (addr < size) v = is_max ? std::max(v, src[addr]) : v + src[addr]; addr += (size_t)1024 * num_wg; }
            sh[t] = v;
        }
        for (uint32_t i = 512; i > 1; i /= 2) for (uint32_t t = 0; t < i; ++t) sh[t] = is_max ? std::max(sh[t], sh[t + i]) : sh[t] + sh[t + i];
        return is_max ? std::max(sh[0], sh[1]) : sh[0] + sh[1];
    }
    float reduce_literal(bool is_max) {
        const uint32_t reduce0_groups = (uint32_t)((N / 16 + 1023) / 1024);    // DispatchCommandReduce0, pressure_init.comp:29
        std::vector<float> a(redbuf), b;
        size_t remaining = N;
        while (remaining > 16384) {                                            // pressure_solver.rs:561-581
            b.assign(std::max<size_t>(reduce0_groups, a.size() / 16384 + 1), 0.0f);
            for (uint32_t w = 0; w < reduce0_groups; ++w) b[w] = reduce_workgroup(a, remaining, w, reduce0_groups, is_max);
            a.swap(b);
            remaining /= 16384;
        }
        return reduce_workgroup(a, remaining, 0, 1, is_max);                   // final dispatch(1,1,1), :586-588
    }

    static inline float eps_div(float num, float den) { return num / (den + (den < 0.0f ? -1e-10f : 1e-10f)); }   // pressure_reduce.comp:71-77

    void solve(int which, float dt) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<float>& p = pressure[which];
        const SolverConfig& c = cfg[which];
        if (!pressure_cleared[which]) { std::fill(p.begin(), p.end(), 0.f); pressure_cleared[which] = true; }   // pressure_solver.rs:601-603
        const float tol = c.error_tolerance / dt;                                                               // :197
        // S0 pressure_init.comp:19-84 (in place on p: only non-FLUID cells are written, only FLUID cells are read)
        {
            std::vector<float> rnew(residual);
#pragma omp parallel for
            for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
                size_t ci = idx(x, y, z);
                if (marker[ci] != CELL_FLUID) continue;
                Nb n = nbm(x, y, z);
                float r = residual[ci];
                float d = 0.f; for (int k = 0; k < 6; ++k) d += std::fabs((float)n.m[k]);
                // NOTE: neighbours that are non-FLUID may already have been zeroed by another invocation; they are
                // only read when FLUID, so the in-place update is race free.
                if (d > 0.0f) r -= d * p[ci];
                if (n.m[0] == CELL_FLUID) r += p[idx(x - 1, y, z)];
                if (n.m[1] == CELL_FLUID) r += p[idx(x + 1, y, z)];
                if (n.m[2] == CELL_FLUID) r += p[idx(x, y - 1, z)];
                if (n.m[3] == CELL_FLUID) r += p[idx(x, y + 1, z)];
                if (n.m[4] == CELL_FLUID) r += p[idx(x, y, z - 1)];
                if (n.m[5] == CELL_FLUID) r += p[idx(x, y, z + 1)];
                rnew[ci] = r;
            }
            residual.swap(rnew);
#pragma omp parallel for
            for (size_t i = 0; i < N; ++i) if (marker[i] != CELL_FLUID) p[i] = 0.f;
        }
        // preconditioner(r) -> s, sigma = s.r   (pressure_solver.rs:636-648, RESULTMODE_INIT)
        precond_pass(residual, aux_temp, false);
        float sigma = (float)precond_pass(aux_temp, search, true);
        float ab = 0.f;
        float max_err = 0.f; int num_iter = 0; bool done = false;
        const int maxit = c.max_num_iterations;
        for (int i = 0; i <= maxit; ++i) {                                    // :654-723
            if (!done) {
                // S4 pressure_apply_coeff.comp:19-30
                std::vector<double> part(nz, 0.0);
                const bool f32dots = dot_mode == 1;
                const bool literal = dot_mode == 2;
                if (literal) redbuf.assign(N, 0.0f);
#pragma omp parallel for
                for (int z = 0; z < nz; ++z) { double acc = 0; float accp = 0.f; for (int y = 0; y < ny; ++y) { float accr = 0.f; for (int x = 0; x < nx; ++x) {
                    size_t ci = idx(x, y, z);
                    if (marker[ci] != CELL_FLUID) continue;
                    float sval = search[ci];
                    const float prod = sval * mulA(search, x, y, z, sval);
                    if (literal) { size_t a = reduce_addr(x, y, z); if (a < N) redbuf[a] = prod; }
                    else if (f32dots) accr += prod; else acc += (double)prod;
                } accp += accr; } part[z] = f32dots ? (double)accp : acc; }
                double dsum = 0;
                if (literal) dsum = reduce_literal(false);
                else if (f32dots) { float fs = 0.f; for (double q : part) fs += (float)q; dsum = fs; } else for (double q : part) dsum += q;
                ab = eps_div(sigma, (float)dsum);                             // RESULTMODE_ALPHA
            }
            const bool check = (i == maxit) || (i > 0 && c.error_check_frequency > 0 && i % c.error_check_frequency == 0);   // :672-673
            float err = 0.f;
            if (!done) {
                // S5 pressure_update_pressure_and_residual.comp:23-59 (A s recomputed from the *old* s: s is not written here)
                std::vector<float> emax(nz, 0.f);
                const bool literal = dot_mode == 2 && check;
                if (literal) redbuf.assign(N, 0.0f);
#pragma omp parallel for
                for (int z = 0; z < nz; ++z) { float e = 0.f; for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
                    size_t ci = idx(x, y, z);
                    if (marker[ci] != CELL_FLUID) continue;
                    float sval = search[ci];
                    p[ci] = p[ci] + ab * sval;
                    float r = residual[ci];
                    r -= ab * mulA(search, x, y, z, sval);
                    residual[ci] = r;
                    e = std::max(e, std::fabs(r));
                    if (literal) { size_t a = reduce_addr(x, y, z); if (a < N) redbuf[a] = std::fabs(r); }
                } emax[z] = e; }
                for (float e : emax) err = std::max(err, e);
                if (literal) err = reduce_literal(true);
            }
            if (check) {
                if (!done && num_iter == 0 && (i == maxit || err < tol)) { max_err = err; num_iter = i; done = true; }   // pressure_reduce.comp:82-94
                if (i == maxit) break;                                                                                    // pressure_solver.rs:695-697
            }
            if (!done) {
                precond_pass(residual, aux_temp, false);
                float sig2 = (float)precond_pass(aux_temp, aux, true);
                ab = eps_div(sig2, sigma); sigma = sig2;                       // RESULTMODE_BETA
                // S6 pressure_update_search.comp:13-24
#pragma omp parallel for
                for (size_t ci = 0; ci < N; ++ci) if (marker[ci] == CELL_FLUID) search[ci] = aux[ci] + ab * search[ci];
            }
        }
        last_stats[which].error = max_err * dt;      // pressure_solver.rs:162
        last_stats[which].iterations = num_iter;
        solver_iterations_total += (uint64_t)num_iter;
        solver_seconds_total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    // ---- D2: divergence_remove.comp:19-49 --------------------------------------------------------
    void divergence_remove() {
        const std::vector<float>& p = pressure[0];
#pragma omp parallel for
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            size_t ci = idx(x, y, z);
            int8_t mc = marker[ci];
            float pc = (mc == CELL_FLUID) ? p[ci] : 0.f;
            for (int c = 0; c < 3; ++c) {
                int n[3] = {x, y, z}; n[c] += 1;
                int8_t mn = mk(n[0], n[1], n[2]);
                float v = 0.f;
                if (mc == CELL_FLUID || mn == CELL_FLUID) {
                    if (mc == CELL_SOLID) v = sv_c(x, y, z, c);
                    else if (mn == CELL_SOLID) v = sv_c(n[0], n[1], n[2], c);
                    else { v = vel[c][ci]; float pn = (mn == CELL_FLUID) ? fv(p, n[0], n[1], n[2]) : 0.f; v -= pc - pn; }
                }
                vel[c][ci] = v;
            }
        }
    }
    // ---- D3: extrapolate_velocity.comp:9-90 ------------------------------------------------------
    void extrapolate_velocity() {
        // Reads only *valid* faces (adjacent to a FLUID cell), writes only invalid ones => in-place is race free.
        static const int OFFS[3][8][3] = {
            {{0,-1,-1},{0,0,-1},{0,1,-1},{0,-1,0},{0,1,0},{0,-1,1},{0,0,1},{0,1,1}},      // :37-44
            {{-1,0,-1},{0,0,-1},{1,0,-1},{-1,0,0},{1,0,0},{-1,0,1},{0,0,1},{1,0,1}},      // :55-62
            {{-1,-1,0},{0,-1,0},{1,-1,0},{-1,0,0},{1,0,0},{-1,1,0},{0,1,0},{1,1,0}}};     // :73-80
#pragma omp parallel for
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            if (marker[idx(x, y, z)] == CELL_FLUID) continue;
            for (int c = 0; c < 3; ++c) {
                int o[3] = {x, y, z}; o[c] += 1;
                if (mk(o[0], o[1], o[2]) == CELL_FLUID) continue;
                float numV = 0.f, avgV = 0.f;
                for (int k = 0; k < 8; ++k) {
                    int cx = x + OFFS[c][k][0], cy = y + OFFS[c][k][1], cz = z + OFFS[c][k][2];
                    int c2[3] = {cx, cy, cz}; c2[c] += 1;
                    bool valid = mk(cx, cy, cz) == CELL_FLUID || mk(c2[0], c2[1], c2[2]) == CELL_FLUID;   // isValidVelocity :9-14
                    if (valid) { numV += 1.f; avgV += fv(vel[c], cx, cy, cz); }
                }
                if (numV > 0.f) vel[c][idx(x, y, z)] = avgV / numV;
            }
        }
    }

    // ---- samplers --------------------------------------------------------------------------------
    inline F4 solid_point_clamp(float tx, float ty, float tz) const {   // SamplerPointClamp on SceneVoxelization
        if (solid.empty()) return F4{0, 0, 0, 0};
        int x = std::min(std::max((int)std::floor(tx * (float)nx), 0), nx - 1);
        int y = std::min(std::max((int)std::floor(ty * (float)ny), 0), ny - 1);
        int z = std::min(std::max((int)std::floor(tz * (float)nz), 0), nz - 1);
        return solid[idx(x, y, z)];
    }
    template <class Fetch> inline float trilinear_clamp(Fetch fetch, float tx, float ty, float tz) const {   // SamplerTrilinearClamp, exact f32 weights
        float ux = tx * (float)nx - 0.5f, uy = ty * (float)ny - 0.5f, uz = tz * (float)nz - 0.5f;
        float fx0 = std::floor(ux), fy0 = std::floor(uy), fz0 = std::floor(uz);
        float fx = ux - fx0, fy = uy - fy0, fz = uz - fz0;
        int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
        auto cl = [](int v, int n) { return std::min(std::max(v, 0), n - 1); };
        int xa = cl(x0, nx), xb = cl(x0 + 1, nx), ya = cl(y0, ny), yb = cl(y0 + 1, ny), za = cl(z0, nz), zb = cl(z0 + 1, nz);
        float c00 = mixf(fetch(xa, ya, za), fetch(xb, ya, za), fx), c10 = mixf(fetch(xa, yb, za), fetch(xb, yb, za), fx);
        float c01 = mixf(fetch(xa, ya, zb), fetch(xb, ya, zb), fx), c11 = mixf(fetch(xa, yb, zb), fetch(xb, yb, zb), fx);
        return mixf(mixf(c00, c10, fy), mixf(c01, c11, fy), fz);
    }

    // shared wall handling: advect_particles.comp:134-173, density_projection_correct_particles.comp:45-69
    inline void truncate_step(const float* orig, const float* move, float* dir, float* max_step) const {
        float len = std::sqrt((move[0] * move[0] + move[1] * move[1]) + move[2] * move[2]) + 1e-10f;
        float ms = len;
        for (int k = 0; k < 3; ++k) {
            dir[k] = move[k] / len;
            float pic = fractf(orig[k]);
            ms = std::min(ms, (dir[k] > 0.0f ? pic : 1.0f - pic) / std::fabs(dir[k]) - 0.001f);   // Q12: literal
        }
        *max_step = ms;
    }

    // ---- A1: advect_particles.comp:35-194 --------------------------------------------------------
    void advect_particles(float dt) {
        const float gs[3] = {(float)nx, (float)ny, (float)nz};
        const float inv[3] = {1.0f / gs[0], 1.0f / gs[1], 1.0f / gs[2]};
        const int dimm1[3] = {nx - 1, ny - 1, nz - 1};
#pragma omp parallel for
        for (int64_t pi = 0; pi < (int64_t)num_particles; ++pi) {
            float op[3] = {pos[pi].x, pos[pi].y, pos[pi].z};
            if (!solid.empty()) {   // :46-65 escape from moving solid
                F4 cs = solid_point_clamp(op[0] * inv[0], op[1] * inv[1], op[2] * inv[2]);
                if (cs.w > 0.0f) {
                    float ax = std::fabs(cs.x), ay = std::fabs(cs.y), az = std::fabs(cs.z);
                    if (ax > ay) { if (ax > az) op[0] += signf(cs.x); else op[2] += signf(cs.z); }
                    else { if (ay > az) op[1] += signf(cs.y); else op[2] += signf(cs.z); }
                }
            }
            float v[8][3];   // corner order 000,100,010,110,001,101,011,111
            float ipx[3], ipy[3], ipz[3];
            for (int i = 0; i < 3; ++i) {   // :74-93
                float off[3] = {0.5f, 0.5f, 0.5f}; off[i] = 1.0f;
                float o[3]; int lo[3], hi[3];
                for (int k = 0; k < 3; ++k) { o[k] = std::max(0.0f, op[k] - off[k]); lo[k] = (int)o[k]; hi[k] = std::min(lo[k] + 1, dimm1[k]); }
                ipx[i] = fractf(o[0]); ipy[i] = fractf(o[1]); ipz[i] = fractf(o[2]);
                const std::vector<float>& V = vel[i];
                v[0][i] = fv(V, lo[0], lo[1], lo[2]); v[1][i] = fv(V, hi[0], lo[1], lo[2]);
                v[2][i] = fv(V, lo[0], hi[1], lo[2]); v[3][i] = fv(V, hi[0], hi[1], lo[2]);
                v[4][i] = fv(V, lo[0], lo[1], hi[2]); v[5][i] = fv(V, hi[0], lo[1], hi[2]);
                v[6][i] = fv(V, lo[0], hi[1], hi[2]); v[7][i] = fv(V, hi[0], hi[1], hi[2]);
            }
            float nv[3], cx[3], cy[3], cz[3];
            for (int i = 0; i < 3; ++i) {   // :97-112 (component-wise vec3 math)
                float x00 = mixf(v[0][i], v[1][i], ipx[i]), x01 = mixf(v[4][i], v[5][i], ipx[i]);
                float x10 = mixf(v[2][i], v[3][i], ipx[i]), x11 = mixf(v[6][i], v[7][i], ipx[i]);
                float xy0 = mixf(x00, x10, ipy[i]), xy1 = mixf(x01, x11, ipy[i]);
                nv[i] = mixf(xy0, xy1, ipz[i]);
                cx[i] = mixf(mixf(v[1][i], v[3][i], ipy[i]), mixf(v[5][i], v[7][i], ipy[i]), ipz[i]) -
                        mixf(mixf(v[0][i], v[2][i], ipy[i]), mixf(v[4][i], v[6][i], ipy[i]), ipz[i]);
                cy[i] = mixf(x10, x11, ipz[i]) - mixf(x00, x01, ipz[i]);
                cz[i] = xy1 - xy0;
            }
            auto tri = [&](const float* sx, const float* sy, const float* sz, float* out) {   // InterpolateTrilinear :21-25
                for (int i = 0; i < 3; ++i)
                    out[i] = mixf(mixf(mixf(v[0][i], v[1][i], sx[i]), mixf(v[2][i], v[3][i], sx[i]), sy[i]),
                                  mixf(mixf(v[4][i], v[5][i], sx[i]), mixf(v[6][i], v[7][i], sx[i]), sy[i]), sz[i]);
            };
            auto shifted = [&](const float* base, const float* step, float* out) { for (int i = 0; i < 3; ++i) out[i] = satf(base[i] + step[i]); };   // Q11: literal
            float k1[3] = {nv[0], nv[1], nv[2]}, k2[3], k3[3], k4[3], st[3], sx[3], sy[3], sz[3];
            for (int i = 0; i < 3; ++i) st[i] = dt * 0.5f * k1[i];                                        // :117
            shifted(ipx, st, sx); shifted(ipy, st, sy); shifted(ipz, st, sz); tri(sx, sy, sz, k2);
            for (int i = 0; i < 3; ++i) st[i] = dt * 0.5f * k2[i];                                        // :120
            shifted(ipx, st, sx); shifted(ipy, st, sy); shifted(ipz, st, sz); tri(sx, sy, sz, k3);
            for (int i = 0; i < 3; ++i) st[i] = dt * k3[i];                                               // :123
            shifted(ipx, st, sx); shifted(ipy, st, sy); shifted(ipz, st, sz); tri(sx, sy, sz, k4);
            float mv[3], np[3];
            for (int i = 0; i < 3; ++i) { mv[i] = dt * (1.0f / 6.0f) * (k1[i] + 2.0f * (k2[i] + k3[i]) + k4[i]); np[i] = op[i] + mv[i]; }   // :126-127
            // :134-173 wall penetration
            bool outside = false;
            for (int k = 0; k < 3; ++k) if (std::min(std::max(np[k], 1.001f), gs[k] - 1.001f) != np[k]) outside = true;
            float tc[3] = {np[0] * inv[0], np[1] * inv[1], np[2] * inv[2]};
            if (outside || (!solid.empty() && solid_point_clamp(tc[0], tc[1], tc[2]).w > 0.0f)) {
                float dir[3], ms; truncate_step(op, mv, dir, &ms);
                for (int k = 0; k < 3; ++k) mv[k] = dir[k] * ms;
                if ((int)op[0] == (int)np[0] && (int)op[1] == (int)np[1] && (int)op[2] == (int)np[2]) {   // :154 stuck anyway
                    auto sw = [&](int x, int y, int z) { return solid.empty() ? 0.f : solid[idx(x, y, z)].w; };
                    float push[3] = {
                        trilinear_clamp(sw, tc[0] - inv[0], tc[1], tc[2]) - trilinear_clamp(sw, tc[0] + inv[0], tc[1], tc[2]),
                        trilinear_clamp(sw, tc[0], tc[1] - inv[1], tc[2]) - trilinear_clamp(sw, tc[0], tc[1] + inv[1], tc[2]),
                        trilinear_clamp(sw, tc[0], tc[1], tc[2] - inv[2]) - trilinear_clamp(sw, tc[0], tc[1], tc[2] + inv[2])};
                    for (int k = 0; k < 3; ++k) mv[k] += push[k] * (dt * 50.0f);
                }
                for (int k = 0; k < 3; ++k) { np[k] = op[k] + mv[k]; np[k] = std::min(std::max(np[k], 1.001f), gs[k] - 1.001f);
                                              nv[k] = (dir[k] * ms) / dt; }   // :166-169
            }
            pos[pi].x = np[0]; pos[pi].y = np[1]; pos[pi].z = np[2];
            pvel[0][pi] = F4{cx[0], cx[1], cx[2], nv[0]};   // :186-188 (Q2: literal layout)
            pvel[1][pi] = F4{cy[0], cy[1], cy[2], nv[1]};
            pvel[2][pi] = F4{cz[0], cz[1], cz[2], nv[2]};
        }
        // :176-181 marker + density linked list (atomic order := ascending index)
        for (uint32_t pi = 0; pi < num_particles; ++pi) {
            PosLl& p = pos[pi];
            int x = (int)p.x, y = (int)p.y, z = (int)p.z;
            if (inb(x, y, z)) marker[idx(x, y, z)] = CELL_FLUID;
            int dx = (int)(p.x - 0.5f), dy = (int)(p.y - 0.5f), dz = (int)(p.z - 0.5f);
            uint32_t old = 0;
            if (inb(dx, dy, dz)) { size_t c = idx(dx, dy, dz); old = ll[c]; ll[c] = pi + 1; }
            p.ll = old - 1u;
        }
    }

    // ---- R1: density_projection_gather_error.comp:41-198 ------------------------------------------
    void density_gather_error(float dt) {
        static const int OFF[8][3] = {{0,0,0},{1,0,0},{0,1,0},{1,1,0},{0,0,1},{1,0,1},{0,1,1},{1,1,1}};
#pragma omp parallel for schedule(dynamic, 1)
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            if (marker[idx(x, y, z)] != CELL_FLUID) continue;   // threadWritesFluid :46
            float sp[3] = {(float)x + 0.5f, (float)y + 0.5f, (float)z + 0.5f};
            float density = 0.f;
            uint32_t cur[8];
            for (int k = 0; k < 8; ++k) { int gx = x - OFF[k][0], gy = y - OFF[k][1], gz = z - OFF[k][2];
                cur[k] = (inb(gx, gy, gz) ? ll[idx(gx, gy, gz)] : 0u) - 1u; }
            for (int round = 0; round < 32; ++round) {   // :69
                bool any = false;
                for (int k = 0; k < 8; ++k) {
                    uint32_t pi = cur[k];
                    if (pi == INVALID_LL) continue;
                    any = true;
                    const PosLl& p = pos[pi]; cur[k] = p.ll;
                    float ox = satf(1.0f - std::fabs(sp[0] - p.x)), oy = satf(1.0f - std::fabs(sp[1] - p.y)), oz = satf(1.0f - std::fabs(sp[2] - p.z));
                    density += ox * oy * oz;   // :27-31
                }
                if (!any) break;
            }
            int8_t m[6] = {mk(x + 1, y, z), mk(x, y + 1, z), mk(x, y, z + 1), mk(x - 1, y, z), mk(x, y - 1, z), mk(x, y, z - 1)};   // :115-120
            bool anyAir = false;
            for (int k = 0; k < 6; ++k) { if (m[k] == CELL_SOLID) density += 0.5625f; if (m[k] == CELL_AIR) anyAir = true; }   // :167-179
            if (anyAir) density = std::max(8.0f, density);   // :182-184
            density = 1.0f - density / 8.0f;                 // :188
            density = std::min(std::max(density, -0.5f), 0.5f);   // :192
            density /= dt;                                   // :196
            residual[idx(x, y, z)] = density;
        }
    }
    // ---- R2: density_projection_position_change.comp:18-51 ----------------------------------------
    void position_change(float dt) {
        const std::vector<float>& p = pressure[1];
#pragma omp parallel for
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            size_t ci = idx(x, y, z);
            int8_t mc = marker[ci];
            float pc = (mc == CELL_FLUID) ? p[ci] : 0.f;
            for (int c = 0; c < 3; ++c) {
                int n[3] = {x, y, z}; n[c] += 1;
                int8_t mn = mk(n[0], n[1], n[2]);
                float pn = (mn == CELL_FLUID) ? fv(p, n[0], n[1], n[2]) : 0.f;
                float d = (pn - pc) * dt;
                if (mc == CELL_SOLID || mn == CELL_SOLID) d = 0.f;
                vel[c][ci] = d;
            }
        }
    }
    // ---- R3: density_projection_correct_particles.comp:25-73 --------------------------------------
    void correct_particles() {
        const float gs[3] = {(float)nx, (float)ny, (float)nz};
        const float inv[3] = {1.0f / gs[0], 1.0f / gs[1], 1.0f / gs[2]};
#pragma omp parallel for
        for (int64_t pi = 0; pi < (int64_t)num_particles; ++pi) {
            float op[3] = {pos[pi].x, pos[pi].y, pos[pi].z};
            float ch[3];
            for (int c = 0; c < 3; ++c) {   // :27-40
                float off[3] = {0.f, 0.f, 0.f}; off[c] = 0.5f;
                float o[3]; for (int k = 0; k < 3; ++k) o[k] = std::max(0.0f, op[k] - off[k]);
                const std::vector<float>& V = vel[c];
                ch[c] = trilinear_clamp([&](int x, int y, int z) { return V[idx(x, y, z)]; }, o[0] * inv[0], o[1] * inv[1], o[2] * inv[2]);
            }
            float np[3] = {op[0] + ch[0], op[1] + ch[1], op[2] + ch[2]};
            bool outside = false;
            for (int k = 0; k < 3; ++k) if (std::min(std::max(np[k], 1.001f), gs[k] - 1.001f) != np[k]) outside = true;
            bool in_solid = false;
            if (!outside) {   // SamplerPointClamp on MarkerVolume :48
                int x = std::min(std::max((int)std::floor(np[0] * inv[0] * gs[0]), 0), nx - 1);
                int y = std::min(std::max((int)std::floor(np[1] * inv[1] * gs[1]), 0), ny - 1);
                int z = std::min(std::max((int)std::floor(np[2] * inv[2] * gs[2]), 0), nz - 1);
                in_solid = marker[idx(x, y, z)] == CELL_SOLID;
            }
            if (outside || in_solid) {
                float dir[3], ms; truncate_step(op, ch, dir, &ms);
                for (int k = 0; k < 3; ++k) { np[k] = op[k] + dir[k] * ms; np[k] = std::min(std::max(np[k], 1.001f), gs[k] - 1.001f); }
            }
            pos[pi].x = np[0]; pos[pi].y = np[1]; pos[pi].z = np[2];
        }
    }
    // ---- B1-B3: particle_binning_{count,prefixsum,rewrite_particles}.comp, hybrid_fluid.rs:854-893 --
    void binning() {
        if (binning_mode == BINNING_OFF) return;
        std::fill(ll.begin(), ll.end(), 0u);                 // clear_texture :858
        const bool literal = binning_mode == BINNING_LITERAL;
        uint32_t T = literal ? std::min<uint32_t>((num_particles + 63) / 64 * 64, max_particles) : num_particles;   // Q4: no i<NumParticles guard
        for (uint32_t i = 0; i < T; ++i) {                   // count :9-13
            PosLl& p = pos[i]; int x = (int)p.x, y = (int)p.y, z = (int)p.z;
            uint32_t old = 0; if (inb(x, y, z)) { old = ll[idx(x, y, z)]; ll[idx(x, y, z)] = old + 1; }
            p.ll = old;
        }
        uint32_t run = 0;                                     // prefixsum :31-61, block order ascending (Appendix A.11)
        for (size_t i = 0; i < N; ++i) { run += ll[i]; if (run != 0) ll[i] = run; }
        for (uint32_t i = 0; i < T; ++i) {                   // rewrite :8-16
            const PosLl& p = pos[i]; int x = (int)p.x, y = (int)p.y, z = (int)p.z;
            uint32_t inc = inb(x, y, z) ? ll[idx(x, y, z)] : 0u;
            uint32_t dst = inc - p.ll - (literal ? 0u : 1u);
            if (dst < max_particles) pos_tmp[dst] = p;
        }
        if (literal) pos = pos_tmp;                           // full-buffer copy :885-891
        else std::copy(pos_tmp.begin(), pos_tmp.begin() + num_particles, pos.begin());
    }

    // ---- stages and HybridFluid::step, hybrid_fluid.rs:770-977 ---------------
    void run_stage(int stage, float dt) {
        switch (stage) {
        case ST_TRANSFER:
            for (int c = 0; c < 3; ++c) { transfer_clear(c); build_linkedlist(c); if (c == 0) set_boundary_marker(); gather_velocity(c, dt); }
            break;
        case ST_DIVERGENCE: divergence_compute(); break;
        case ST_SOLVE_VELOCITY: solve(0, dt); break;
        case ST_BINNING: binning(); break;
        case ST_PROJECT: divergence_remove(); extrapolate_velocity(); break;
        case ST_ADVECT: transfer_clear(0); advect_particles(dt); set_boundary_marker(); break;
        case ST_DENSITY_GATHER: density_gather_error(dt); break;
        case ST_SOLVE_DENSITY: solve(1, dt); break;
        case ST_POSITION_CHANGE: position_change(dt); extrapolate_velocity(); break;
        case ST_CORRECT: correct_particles(); break;
        }
    }
    void step(float dt) {
        run_stage(ST_TRANSFER, dt); run_stage(ST_DIVERGENCE, dt); run_stage(ST_SOLVE_VELOCITY, dt);
        if (rebin_freq != 0 && step_counter % rebin_freq == 0) run_stage(ST_BINNING, dt);   // :854-856 (Q13)
        run_stage(ST_PROJECT, dt); run_stage(ST_ADVECT, dt); run_stage(ST_DENSITY_GATHER, dt);
        run_stage(ST_SOLVE_DENSITY, dt); run_stage(ST_POSITION_CHANGE, dt); run_stage(ST_CORRECT, dt);
        step_counter += 1;
    }

    // ---- static objects: scene/voxelization.rs:116-157 + shader/voxelize/conservative_hull.{vert,frag} -----------------------
    // Sequential restatement of the conservative-hull pass: vertex stage, an "overestimate" conservative rasteriser (a pixel
    // yields a fragment iff its unit square overlaps the projected triangle), fragment stage.  What Vulkan leaves to the
    // implementation is fixed as documented in include/blubhip.h (no sub-pixel snapping, depth = plane at the pixel centre
    // clamped to the triangle's range, plane slopes for dFdx/dFdyCoarse, RGBA16F stores round to nearest even).  PARITY
    // UNPINNED like the rest of the oracle: no reference test or fixture covers the voxelisation.
    struct MeshDesc { float m[3][4]; float vel[3]; float axis[3]; uint32_t index_begin, index_end; };
    static float f16_round(float v) {   // f32 -> f16 (round to nearest even) -> f32
        uint32_t u; memcpy(&u, &v, 4);
        const uint32_t sign = u & 0x80000000u; u &= 0x7FFFFFFFu;
        if (u >= 0x7F800000u) return v;                                                   // inf / nan
        if (u >= 0x477FF000u) { uint32_t inf = sign | 0x7F800000u; float r; memcpy(&r, &inf, 4); return r; }   // >= 65520 rounds to inf
        if (u < 0x38800000u) {                                                            // f16 subnormal range: quantum 2^-24
            float a; memcpy(&a, &u, 4);
            const float q = std::nearbyint(a * 16777216.0f) / 16777216.0f;                // default rounding mode = nearest even
            uint32_t r; memcpy(&r, &q, 4); r |= sign; float out; memcpy(&out, &r, 4); return out;
        }
        const uint32_t rem = u & 0x1FFFu;                                                 // 13 dropped mantissa bits
        u &= ~0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (u & 0x2000u))) u += 0x2000u;
        u |= sign; float out; memcpy(&out, &u, 4); return out;
    }
    void frag_store(const MeshDesc& d, const float vp[3]) {   // ComputeVoxelSpeed + imageStore (conservative_hull.frag:20-26)
        const float px = vp[0] - d.m[0][3], py = vp[1] - d.m[1][3], pz = vp[2] - d.m[2][3];
        const float dt = px * d.axis[0] + py * d.axis[1] + pz * d.axis[2];
        const float tx = px - dt * d.axis[0], ty = py - dt * d.axis[1], tz = pz - dt * d.axis[2];
        const float vx = (d.axis[1] * tz - d.axis[2] * ty) + d.vel[0];
        const float vy = (d.axis[2] * tx - d.axis[0] * tz) + d.vel[1];
        const float vz = (d.axis[0] * ty - d.axis[1] * tx) + d.vel[2];
        const int ix = (int)vp[0], iy = (int)vp[1], iz = (int)vp[2];
        if (inb(ix, iy, iz)) solid[idx(ix, iy, iz)] = F4{f16_round(vx), f16_round(vy), f16_round(vz), 1.0f};
    }
    void unswizzle_clamp(int side, float sx, float sy, float sz, float out[3]) const {   // :17-18
        float x, y, z;
        if (side == 0) { x = sz; y = sy; z = sx; } else if (side == 1) { x = sx; y = sz; z = sy; } else { x = sx; y = sy; z = sz; }
        out[0] = std::fmin(std::fmax(x, 0.0f), (float)nx - 1.0f);
        out[1] = std::fmin(std::fmax(y, 0.0f), (float)ny - 1.0f);
        out[2] = std::fmin(std::fmax(z, 0.0f), (float)nz - 1.0f);
    }
    void voxelize(const float* positions, const uint32_t* indices, uint32_t num_meshes, const MeshDesc* meshes) {
        solid.assign(N, F4{0, 0, 0, 0});   // clear_texture, voxelization.rs:123
        const float viewport = (float)std::max(nx, std::max(ny, nz));   // :99
        for (uint32_t mi = 0; mi < num_meshes; ++mi) {
            const MeshDesc& d = meshes[mi];
            for (uint32_t first = d.index_begin; first + 3 <= d.index_end; first += 3) {
                // vertex stage (conservative_hull.vert:14-33)
                float v[3][3];
                for (int k = 0; k < 3; ++k) {
                    const float* p = positions + 3 * (size_t)indices[first + k];
                    for (int r = 0; r < 3; ++r) v[k][r] = p[0] * d.m[r][0] + p[1] * d.m[r][1] + p[2] * d.m[r][2] + d.m[r][3];
                }
                const float e1x = v[1][0] - v[0][0], e1y = v[1][1] - v[0][1], e1z = v[1][2] - v[0][2];
                const float e2x = v[2][0] - v[0][0], e2y = v[2][1] - v[0][1], e2z = v[2][2] - v[0][2];
                const float nax = std::fabs(e1y * e2z - e1z * e2y), nay = std::fabs(e1z * e2x - e1x * e2z), naz = std::fabs(e1x * e2y - e1y * e2x);
                int side = nax > nay ? 0 : 1;
                side = (side == 0 ? nax : nay) > naz ? side : 2;
                float s[3][3];   // swizzled window-space vertices
                for (int k = 0; k < 3; ++k) {
                    if (side == 0) { s[k][0] = v[k][2]; s[k][1] = v[k][1]; s[k][2] = v[k][0]; }
                    else if (side == 1) { s[k][0] = v[k][0]; s[k][1] = v[k][2]; s[k][2] = v[k][1]; }
                    else { s[k][0] = v[k][0]; s[k][1] = v[k][1]; s[k][2] = v[k][2]; }
                }
                // rasteriser: cull_mode None => both windings; degenerate projections produce nothing
                float area2 = (s[1][0] - s[0][0]) * (s[2][1] - s[0][1]) - (s[1][1] - s[0][1]) * (s[2][0] - s[0][0]);
                if (!(area2 != 0.0f) || !(std::fabs(area2) < 3.0e38f)) continue;
                if (area2 < 0.0f) { for (int c = 0; c < 3; ++c) std::swap(s[1][c], s[2][c]); area2 = -area2; }
                const float ax = s[0][0], ay = s[0][1], az = s[0][2], bx = s[1][0], by = s[1][1], bz = s[1][2], cx = s[2][0], cy = s[2][1], cz = s[2][2];
                const float dzdx = ((bz - az) * (cy - ay) - (cz - az) * (by - ay)) / area2;
                const float dzdy = ((cz - az) * (bx - ax) - (bz - az) * (cx - ax)) / area2;
                const float zmin = std::fmin(az, std::fmin(bz, cz)), zmax = std::fmax(az, std::fmax(bz, cz));
                const float xlo = std::fmax(std::floor(std::fmin(ax, std::fmin(bx, cx))), 0.0f), xhi = std::fmin(std::floor(std::fmax(ax, std::fmax(bx, cx))), viewport - 1.0f);
                const float ylo = std::fmax(std::floor(std::fmin(ay, std::fmin(by, cy))), 0.0f), yhi = std::fmin(std::floor(std::fmax(ay, std::fmax(by, cy))), viewport - 1.0f);
                if (!(xhi >= xlo) || !(yhi >= ylo)) continue;
                const float ex[3] = {bx - ax, cx - bx, ax - cx}, ey[3] = {by - ay, cy - by, ay - cy};
                const float ox[3] = {ax, bx, cx}, oy[3] = {ay, by, cy};
                for (int j = (int)ylo; j <= (int)yhi; ++j)
                    for (int i = (int)xlo; i <= (int)xhi; ++i) {
                        bool covered = true;
                        for (int e = 0; e < 3 && covered; ++e) {   // the corner of the pixel square that maximises the edge function
                            const float qx = ey[e] < 0.0f ? (float)i + 1.0f : (float)i, qy = ex[e] > 0.0f ? (float)j + 1.0f : (float)j;
                            covered = (ex[e] * (qy - oy[e]) - ey[e] * (qx - ox[e])) >= 0.0f;
                        }
                        if (!covered) continue;
                        // fragment stage (conservative_hull.frag:28-56); gl_FragCoord = (i + .5, j + .5, depth)
                        const float fcx = (float)i + 0.5f, fcy = (float)j + 0.5f;
                        float z = az + dzdx * (fcx - ax) + dzdy * (fcy - ay);
                        z = std::fmin(std::fmax(z, zmin), zmax);
                        if (!(z >= 0.0f && z <= viewport)) continue;   // depth clipping
                        float vp[3];
                        unswizzle_clamp(side, std::trunc(fcx), std::trunc(fcy), std::trunc(z), vp);   // ivec3(voxelPosSwizzled), :35-36
                        frag_store(d, vp);
                        const float max_change = std::fmax(std::fabs(dzdx), std::fabs(dzdy));            // :39-44
                        if (std::floor(z) != std::floor(z - max_change)) { unswizzle_clamp(side, fcx, fcy, z - 1.0f, vp); frag_store(d, vp); }   // :46-49
                        if (std::floor(z) != std::floor(z + max_change)) { unswizzle_clamp(side, fcx, fcy, z + 1.0f, vp); frag_store(d, vp); }   // :50-53
                    }
            }
        }
    }

    void* volume_ptr(int which, size_t* bytes) {
        switch (which) {
        case V_MARKER: *bytes = N; return marker.data();
        case V_LL: *bytes = N * 4; return ll.data();
        case V_VELX: case V_VELY: case V_VELZ: *bytes = N * 4; return vel[which - V_VELX].data();
        case V_PRESSURE_VELOCITY: case V_PRESSURE_DENSITY: *bytes = N * 4; return pressure[which - V_PRESSURE_VELOCITY].data();
        case V_RESIDUAL: *bytes = N * 4; return residual.data();
        case V_SEARCH: *bytes = N * 4; return search.data();
        case V_AUX: *bytes = N * 4; return aux.data();
        case V_AUX_TEMP: *bytes = N * 4; return aux_temp.data();
        case V_SOLID: *bytes = N * 16; return solid.empty() ? nullptr : solid.data();
        }
        *bytes = 0; return nullptr;
    }
};
}  // namespace

extern "C" {
void* orc_create(int nx, int ny, int nz, uint32_t max_particles) { return new Oracle(nx, ny, nz, max_particles); }
void orc_destroy(void* h) { delete (Oracle*)h; }
int orc_add_fluid_cube(void* h, const float* mn, const float* mx) { return ((Oracle*)h)->add_fluid_cube(mn, mx); }
void orc_set_gravity_grid(void* h, const float* g) { for (int k = 0; k < 3; ++k) ((Oracle*)h)->gravity[k] = g[k]; }
void orc_set_solver_config(void* h, int which, float tol, int max_iter, int freq) { ((Oracle*)h)->cfg[which] = SolverConfig{tol, max_iter, freq}; }
void orc_set_quirks(void* h, int precond_mode, int binning_mode) { ((Oracle*)h)->precond_mode = precond_mode; ((Oracle*)h)->binning_mode = binning_mode; }
void orc_set_rebinning_frequency(void* h, uint32_t f) { ((Oracle*)h)->rebin_freq = f; }
void orc_set_dot_mode(void* h, int mode) { ((Oracle*)h)->dot_mode = mode; }
void orc_reset_pressure_cleared(void* h, int which, int cleared) { ((Oracle*)h)->pressure_cleared[which] = cleared != 0; }
uint32_t orc_num_particles(void* h) { return ((Oracle*)h)->num_particles; }
uint32_t orc_step_counter(void* h) { return ((Oracle*)h)->step_counter; }
void orc_set_step_counter(void* h, uint32_t c) { ((Oracle*)h)->step_counter = c; }
int orc_set_particles(void* h, uint32_t n, const void* pos_ll, const void* vx, const void* vy, const void* vz) {
    Oracle* o = (Oracle*)h; if (n > o->max_particles) return -1;
    o->num_particles = n;
    if (pos_ll) memcpy(o->pos.data(), pos_ll, (size_t)n * 16);
    const void* src[3] = {vx, vy, vz};
    for (int c = 0; c < 3; ++c) { if (src[c]) memcpy(o->pvel[c].data(), src[c], (size_t)n * 16); else std::fill(o->pvel[c].begin(), o->pvel[c].begin() + n, F4{0, 0, 0, 0}); }
    return 0;
}
void orc_get_particles(void* h, void* pos_ll, void* vx, void* vy, void* vz) {
    Oracle* o = (Oracle*)h; size_t b = (size_t)o->num_particles * 16;
    if (pos_ll) memcpy(pos_ll, o->pos.data(), b);
    void* dst[3] = {vx, vy, vz};
    for (int c = 0; c < 3; ++c) if (dst[c]) memcpy(dst[c], o->pvel[c].data(), b);
}
int orc_read_volume(void* h, int which, void* out) { size_t b; void* p = ((Oracle*)h)->volume_ptr(which, &b); if (!p) return -1; memcpy(out, p, b); return 0; }
int orc_write_volume(void* h, int which, const void* in) {
    Oracle* o = (Oracle*)h;
    if (which == V_SOLID) { if (!in) { o->solid.clear(); return 0; } o->solid.resize(o->N); }
    size_t b; void* p = o->volume_ptr(which, &b); if (!p) return -1; memcpy(p, in, b); return 0;
}
int orc_voxelize(void* h, const float* positions, const uint32_t* indices, uint32_t num_meshes, const void* mesh_descs) {
    static_assert(sizeof(Oracle::MeshDesc) == 80, "layout of blub_mesh_desc");
    ((Oracle*)h)->voxelize(positions, indices, num_meshes, (const Oracle::MeshDesc*)mesh_descs);
    return 0;
}
float orc_f16_round(float v) { return Oracle::f16_round(v); }
void orc_run_stage(void* h, int stage, float dt) { ((Oracle*)h)->run_stage(stage, dt); }
void orc_step(void* h, float dt) { ((Oracle*)h)->step(dt); }
void orc_get_solver_stats(void* h, int which, float* err, int* it) { *err = ((Oracle*)h)->last_stats[which].error; *it = ((Oracle*)h)->last_stats[which].iterations; }
void orc_get_solver_totals(void* h, uint64_t* iters, double* seconds) { *iters = ((Oracle*)h)->solver_iterations_total; *seconds = ((Oracle*)h)->solver_seconds_total; }
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
    (void)n;
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
// RNG known-answer hooks (tests/test_oracle_kat.py)
void orc_rng_from_seed(const uint8_t* seed32, uint64_t* out, int n) { Oracle::SmallRng r; r.from_seed_bytes(seed32); for (int i = 0; i < n; ++i) out[i] = r.next_u64(); }
void orc_rng_seed_from_u64(uint64_t seed, uint64_t* state4, float* out, int n) { Oracle::SmallRng r; r.seed_from_u64(seed); for (int k = 0; k < 4; ++k) state4[k] = r.s[k]; for (int i = 0; i < n; ++i) out[i] = r.gen_f32(); }
}
