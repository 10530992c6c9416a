// libblubhip.so -- host orchestration + C-ABI of the MI355X-native blub fluid step.
// Mirrors src/simulation/hybrid_fluid.rs (HybridFluid) and src/simulation/pressure_solver.rs (PressureSolver /
// PressureField) of the reference; the kernels live in blub_kernels.hip.h.  There is no CPU fallback: without a HIP
// device every device entry point returns BLUB_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "blub_internal.h"
#include "blub_pcg_dense.hip.h"
#include "blub_pcg1.hip.h"
#include "blub_slab.hip.h"
#include "blub_voxelize.hip.h"

namespace blub {

static thread_local std::string g_last_error;
int set_error(int status, const char* msg) { g_last_error = msg ? msg : ""; return status; }

#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t _e = (expr);                                                                                         \
        if (_e != hipSuccess) {                                                                                         \
            char _b[512]; snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return set_error(_e == hipErrorOutOfMemory ? BLUB_ERR_OUT_OF_MEMORY : (_e == hipErrorNoDevice ? BLUB_ERR_NO_DEVICE : BLUB_ERR_DEVICE), _b); \
        }                                                                                                               \
    } while (0)

using namespace blubk;

enum KernelClass {
    KC_BRICK_LISTS, KC_RESET_BRICKS, KC_BUILD_LISTS, KC_GATHER_VELOCITY, KC_DIVERGENCE, KC_PCG_INIT, KC_PCG_DIR, KC_PCG_UPDATE, KC_PCG_FINALIZE,
    KC_PCG_ITER, KC_PCG_LOD0, KC_DIVERGENCE_REMOVE, KC_EXTRAPOLATE, KC_ADVECT, KC_DENSITY_GATHER, KC_POSITION_CHANGE, KC_CORRECT,
    KC_BIN_COUNT, KC_BIN_SCAN, KC_BIN_REWRITE, KC_COPY, KC_VOXELIZE, KC_COUNT
};
static const char* kKernelClassNames[KC_COUNT] = {
    "brick_lists", "reset_bricks", "build_lists", "gather_velocity", "divergence", "pcg_init", "pcg_dir", "pcg_update", "pcg_finalize",
    "pcg_iter", "pcg_lod0", "divergence_remove", "extrapolate", "advect", "density_gather", "position_change", "correct",
    "bin_count", "bin_scan", "bin_rewrite", "copy", "voxelize"};

constexpr int PCG_GRID_MAX = 4096;   // upper bound of persistent blocks of the PCG kernels (= number of dot-product partials)
constexpr int PCG_GRID_DENSE = 2048; // dense rows: 8 blocks of 256 threads per CU
constexpr int BRICK_GRID_MAX = 2048; // persistent blocks of the brick-list kernels
constexpr int STATS_RING = 32;       // pressure_solver.rs:49 NUM_PRESSURE_ERROR_BUFFER
constexpr size_t STATS_HISTORY = 100;   // pressure_solver.rs:101
constexpr int COUNTS_RING = 32;
constexpr float SPARSE_PCG_MAX_FILL = 0.30f;   // fluid bricks / bricks below which the brick-list PCG kernels are used (single-reduction schedule; see stage_solve)
constexpr float SPARSE_PCG_MAX_FILL_REFERENCE = 0.20f;   // ... with the reference's two-reduction order

struct PendingStat { uint32_t seq; int slot; };

}  // namespace blub
using namespace blub;

struct blub_fluid {
    Grid g{};
    size_t N = 0;
    // Planes of the grid the volumes hold: [vol_z0, vol_z0 + vol_planes).  Everything for a single domain; a z-slab of a group allocates only its own
    // planes plus BZ on either side.  The volume POINTERS stay those of plane 0 (allocation - vol_first elements), so kernels index globally.
    int vol_z0 = 0, vol_planes = 0; size_t vol_cells = 0, vol_first = 0;
    std::vector<void*> vol_owned;   // volumes allocated one by one (no volume slab)
    float4* solid_alloc = nullptr;   // (solid = solid_alloc - vol_first)
    int mem_mode = 0;                // BLUB_SLAB_MEMORY_*: how the volume slab and the recurrence volumes are allocated (slabs of a group only)
    float* cgbuf_alloc[3] = {nullptr, nullptr, nullptr};   // (cgbuf[] rotates with residual / search, which live in the volume slab)
    uint32_t max_particles = 0, num_particles = 0;
    uint32_t last_add_dropped = 0;   // particles the last add_fluid_cube could not add (capacity)
    // z-slab decomposition (blub_slab.hip): own planes [slab_z0, slab_z1), ghost particles live at [num_particles, +num_ghost)
    int slab_z0 = 0, slab_z1 = 0;
    uint32_t num_ghost = 0;
    // z-slab groups without host synchronisation (blub_slab.inc.hip): the particle counts live on the device ({own, ghost}); num_particles /
    // num_ghost are then the BOUNDS launch grids are sized for, and the exact values only come back on request (slab_refresh_counts)
    uint32_t* n_dev = nullptr;
    // internal re-sort of the particles (blub_bricks.hip.h "internal re-sort"): every `resort_every` steps, at the binning point of blub_fluid_step
    int resort_every = 8;                     // blub_fluid_set_tuning "resort_every": 0 = never (the particle order only changes with the reference's rebinning)
    uint32_t *pid = nullptr, *pid_tmp = nullptr;          // internal slot -> canonical particle index (allocated with the tables, on the first re-sort)
    uint32_t *resort_counters = nullptr, *resort_starts = nullptr, *resort_ranks = nullptr;   // brick-major tables (nb x 512 + 4 entries), a rank per particle
    uint32_t* resort_cursor = nullptr;        // bump allocator of k_resort_scan (zeroed by every brick list build)
    bool order_internal = false;              // the particle arrays are NOT in the caller's order (pid is not the identity)
    uint32_t resorts_done = 0;
    bool bricks_premarked = false;            // brick_fluid already holds the marks of the current particle positions (set and consumed inside stage_advect)
    float gravity[3] = {0, 0, 0};
    int device = 0;
    char* slab = nullptr; size_t slab_bytes = 0, slab_used = 0, slab_shift = 0; int slab_count = 0;   // grid volumes from one allocation (see vol_alloc)
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    uint32_t precond_mode = BLUB_PRECOND_ZERO, binning_mode = BLUB_BINNING_FIXED;
    int filter_mode = BLUB_FILTER_SEPARABLE;   // arithmetic of the trilinear filter in R3 and in A1's push-out (blub_fluid_set_filter_mode)
    uint32_t rebin_freq = 60;   // hybrid_fluid.rs:603-605
    uint32_t step_counter = 0;
    // particles (hybrid_fluid.rs:114-122)
    float4 *pos = nullptr, *pos_tmp = nullptr, *pvel[3] = {nullptr, nullptr, nullptr};
    blubk::GatherNode* nodes = nullptr;       // 3 x 32 bytes per particle: what the P2G list walks read (written by k_build_lists); component c at nodes + c * node_stride
    uint32_t node_stride = 0;
    // volumes (hybrid_fluid.rs:142-154; pressure_solver.rs:104-108, 332-351)
    int8_t* marker = nullptr;
    uint32_t* ll[3] = {nullptr, nullptr, nullptr};
    float *vel[3] = {nullptr, nullptr, nullptr}, *pressure[2] = {nullptr, nullptr}, *residual = nullptr, *search = nullptr, *aux = nullptr, *aux_temp = nullptr;
    float4* solid = nullptr;
    // static object meshes (scene/models.rs:354-375): positions (3 floats per vertex) + triangle indices
    float* mesh_positions = nullptr;
    uint32_t* mesh_indices = nullptr;
    uint32_t mesh_num_vertices = 0, mesh_num_indices = 0;
    uint32_t* scan_totals = nullptr;
    // brick work lists (blub_bricks.hip.h)
    BrickGeom bg{};
    int brick_grid = 0;
    uint8_t *brick_fluid = nullptr, *brick_active = nullptr, *brick_touched = nullptr;
    uint32_t *list_fluid = nullptr, *list_active = nullptr, *list_reset = nullptr;
    uint8_t* brick_flags = nullptr;
    uint4* brick_block_counts = nullptr;
    uint32_t* brick_block_ready = nullptr;    // per block of k_bricks_build: sequence number of the last build it has classified
    int num_cus = 0;
    BrickCounts* counts = nullptr;            // device
    BrickCounts last_counts{}; bool have_last_counts = false;      // newest snapshot latest_counts() has seen land
    BrickCounts* counts_host = nullptr;       // pinned ring of COUNTS_RING snapshots (path selection only), tagged by seq
    BrickCounts* counts_host_dev = nullptr;   // the same ring as the device sees it (kernels write the snapshots directly)
    uint32_t counts_seq = 0;                  // number of list builds enqueued so far
    // run-ahead throttle: the device writes the number of the last finished step into pinned host memory
    volatile uint32_t* steps_done_host = nullptr;
    uint32_t* steps_done_dev = nullptr;
    uint32_t steps_enqueued = 0;
    uint32_t max_steps_in_flight = 4;
    bool all_touched = false;
    int force_pcg_path = -1;                  // -1 auto, 0 dense rows, >= 1 brick lists
    int dense_kd_nt = -1;                     // non-temporal s_out stores of the dense direction kernel: -1 = by grid size, 0 / 1 = forced (tuning knob)
    int fuse_divergence = 1; bool divergence_deferred = false;     // ("fuse_divergence" tuning: 0 never, 1 inside blub_fluid_step, 2 also for blub_fluid_run_stage -- a test hook; see stage_divergence)
    int p2g_own = 1;                          // P2G gather of a single domain: 1 = list-centric (every list walked once + k_gather_finish3; blub_bricks.hip.h), 0 = tile-centric (halo lists walked by every brick that needs them); "p2g_own" tuning
    float2* gather_sums = nullptr; uint32_t* gather_stamp = nullptr;      // list-centric gather: region sums / stamps per component and brick (allocated on first use)
    int p2g_compact = -1;                     // P2G gather: 1 = the tile's non-empty lists compacted (k_gather_velocity3_s), 0 = one lane per list cell, -1 = by fill (stage_transfer)
    bool two_kernel_build = false;            // test hook ("bricks_two_kernel_build"): the list build of grids with more brick blocks than CUs
    int dense_alternate_march = -1;           // odd z-chunks of the dense kernels march downwards (blub_pcg_dense.hip.h: xcd_tile_pairs): bit 0 KD, bit 1 KU, -1 by grid size
    bool standalone_stage = false;            // the running stage was called through blub_fluid_run_stage (test hook)
    int list_grid_forced = 0;                 // test hook ("list_launch_grid"): launch grid of the brick-list kernels
    int pcg_grid_forced = 0;                  // test hook (blub_fluid_set_tuning "pcg_launch_grid"): launch grid of the brick-mapped PCG kernels, 0 = estimated
    // PCG
    uint8_t* dvol = nullptr;
    PcgGeom geom{};
    int pcg_grid = 0;
    PcgGeomZ gz{};            // 2.5-D dense mapping (blub_pcg_dense.hip.h)
    PcgGeomZ gzd{};           // ... as the direction kernel sees it: `dense_kd_chunk_factor` chunks of gz per tile
    int dense_kd_chunk_factor = 0;   // 0 = by grid size (set_dense_geometry)
    int pcg_grid_z = 0;
    float *part_sas = nullptr, *part_sigma[2] = {nullptr, nullptr}, *part_max = nullptr;
    uint8_t* tile_flags = nullptr;
    PcgCtrl* ctrl[2] = {nullptr, nullptr};
    // single-reduction schedule (blub_pcg1.hip.h): second buffers of r / w / q (allocated on first use), float4 partials, scalars
    int pcg_schedule = 1;            // 1 (default since round 4): one kernel per iteration on the brick mapping (single-reduction form of the same recurrence, include/blubhip.h);
                                     // 0: the reference's two-reduction order (pressure_solver.rs:654-723), also taken by solves configured beyond pcg1_max_iterations
    float* cgbuf[3] = {nullptr, nullptr, nullptr};
    float4* part4 = nullptr;
    Pcg1Scalars* pcg1_scalars[2] = {nullptr, nullptr};
    unsigned long long* phase_stamps[2] = {nullptr, nullptr};   // diagnostic ("pcg_phase_stamps" tuning): 64 x 8 time stamps per solver, see SlabDirect::stamps / blub_fluid_read_phase_stamps
    float4* scalar_log[2] = {nullptr, nullptr};   // diagnostic ("pcg_scalar_log" tuning): 1024 entries per solver, see SlabDirect::log / blub_fluid_read_scalar_log
    PcgTailSync* tail_sync[2] = {nullptr, nullptr};
    bool use_tail = true;            // persistent tail kernel of the single-reduction solves (blub_fluid_set_tuning "pcg_tail")
    int tail_margin_checks = 1;
    int pcg1_max_iterations = 64;    // solves with more iterations run the reference order even when schedule 1 is selected (drift of the recurrence residual)
    int tail_grid = 256, tail_grid_max = 256;   // co-resident blocks of the tail kernel: occupancy x CUs (set at creation)
    bool tail_inject_timeout = false;   // test hook (blub_fluid_set_tuning "pcg_tail_inject_timeout"): the next tail kernel finds its grid barrier timed out
    int tail_first_forced = -1;      // test hook (blub_fluid_set_tuning "pcg_tail_first"): hand over to the tail after exactly this many launched iterations
    blub_solver_config cfg[2] = {{0.1f, 32, 4}, {0.1f, 32, 4}};   // hybrid_fluid.rs:253-257
    bool pressure_initialised[2] = {false, false};
    int last_schedule[2] = {-1, -1}, last_mapping[2] = {-1, -1};   // what the most recently enqueued solve actually ran (blub_fluid_last_solve_path)
    // statistics read-back ring (pressure_solver.rs:118-126, 148-209)
    PcgCtrl* stats_host[2] = {nullptr, nullptr};   // pinned ring of STATS_RING control-block snapshots, tagged by seq
    PcgCtrl* stats_host_dev[2] = {nullptr, nullptr};
    uint32_t solve_seq[2] = {0, 0};                // number of solves enqueued so far
    std::deque<PendingStat> stats_pending[2];
    std::deque<float> stats_dt[2];
    std::deque<blub_solver_stats> stats_history[2];
    uint64_t total_iterations = 0;
    uint32_t failed_solves = 0, failed_solves_reported = 0;   // solves whose tail kernel reported a grid-barrier timeout (num_iter < 0)
    // profiling
    bool prof_enabled = false;
    struct ProfPending { hipEvent_t a, b; int kc; int stage; uint32_t step; };
    std::vector<ProfPending> prof_pending;
    struct TraceEvent { int kc; int stage; uint32_t step; double start_us, dur_us; };
    std::vector<TraceEvent> prof_trace;          // bounded per-launch timeline
    hipEvent_t prof_origin = nullptr;            // first profiled launch since reset
    int cur_stage = BLUB_STAGE_COUNT;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[KC_COUNT] = {};
    uint64_t prof_launches[KC_COUNT] = {};
};

namespace blub {

static int prof_flush(blub_fluid* h) {
    if (h->prof_pending.empty()) return BLUB_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& p : h->prof_pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        h->prof_ms[p.kc] += ms; h->prof_launches[p.kc] += 1;
        if (h->prof_trace.size() < 65536) {
            if (!h->prof_origin) { h->prof_origin = p.a; }
            float since = 0.f;
            if (p.a != h->prof_origin) HIP_TRY(hipEventElapsedTime(&since, h->prof_origin, p.a));
            h->prof_trace.push_back({p.kc, p.stage, p.step, since * 1e3, ms * 1e3});
        }
        if (p.a != h->prof_origin) h->prof_pool.push_back(p.a);
        h->prof_pool.push_back(p.b);
    }
    h->prof_pending.clear();
    return BLUB_OK;
}
struct ProfScope {
    blub_fluid* h; int kc; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(blub_fluid* h_, int kc_) : h(h_), kc(kc_) {
        if (!h->prof_enabled) return;
        if (h->prof_pending.size() >= 8192) prof_flush(h);
        auto get = [&]() { hipEvent_t e; if (!h->prof_pool.empty()) { e = h->prof_pool.back(); h->prof_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, h->stream);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, h->stream);
        h->prof_pending.push_back({a, b, kc, h->cur_stage, h->step_counter});
    }
};
// Profiled kernel launches carry their events INSIDE the dispatch (hipExtLaunchKernelGGL: the events take the kernel's own start / end
// time stamps), so the reported durations are the kernels' -- an event pair recorded around a launch adds ~2 us of its own to each of them
// (round 2: the classes summed to more than the step; round-2 review).  Copies / memsets keep the recorded pair (ProfScope).
static void prof_events(blub_fluid* h, hipEvent_t* a, hipEvent_t* b) {
    if (h->prof_pending.size() >= 8192) prof_flush(h);
    auto get = [&]() { hipEvent_t e; if (!h->prof_pool.empty()) { e = h->prof_pool.back(); h->prof_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    *a = get(); *b = get();
}
#define LAUNCH_LDS(h, kc, kernel, grid, block, lds_bytes, ...)                                                         \
    do {                                                                                                               \
        if ((h)->prof_enabled) {                                                                                       \
            hipEvent_t _a, _b; prof_events((h), &_a, &_b);                                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, (lds_bytes), (h)->stream, _a, _b, 0, __VA_ARGS__);              \
            (h)->prof_pending.push_back({_a, _b, (kc), (h)->cur_stage, (h)->step_counter});                            \
        } else hipLaunchKernelGGL(kernel, grid, block, (lds_bytes), (h)->stream, __VA_ARGS__);                         \
    } while (0)
#define LAUNCH(h, kc, kernel, grid, block, ...) LAUNCH_LDS(h, kc, kernel, grid, block, 0, __VA_ARGS__)

static unsigned particle_blocks(uint32_t n) { return (n + 255) / 256; }
[[maybe_unused]] constexpr uint32_t N_OWN = 1u, N_GHOST = 2u, N_ALL = 3u;      // particle_count() selectors (blub_kernels.hip.h)
static unsigned stream_blocks(size_t items) { return (unsigned)std::min<size_t>((items + 255) / 256, 2048); }

// NOTE: the handle's stream is non-blocking, i.e. NOT ordered against the null stream: every memset / copy of this
// library is issued on the handle's stream (a null-stream hipMemset may still be in flight when a kernel starts).
// Memory that another agent writes while kernels of this process run (the direct transport of a z-slab group, include/blubhip.h: BLUB_SLAB_MEMORY_*)
static hipError_t shared_malloc(void** p, size_t bytes, int mem_mode) {
    if (mem_mode == BLUB_SLAB_MEMORY_FINE_GRAINED) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    if (mem_mode == BLUB_SLAB_MEMORY_UNCACHED) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    return hipMalloc(p, bytes);
}
template <class T>
static int dev_alloc_zero(hipStream_t stream, T** p, size_t count) {
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    HIP_TRY(hipMemsetAsync(*p, 0, count * sizeof(T), stream));
    return BLUB_OK;
}
// Grid volumes are carved from ONE allocation, volume i shifted by i x 64 KiB against its 2 MiB-aligned slot.  Measured at 512^3
// (profiles/r03_volume_placement_512.txt): with one hipMalloc per volume -- all 2 MiB aligned, i.e. every concurrent stream of a stencil
// kernel (r, s, p, descriptor at the same in-plane offset) starts in the same channel / bank phase -- the dense PCG update kernel takes
// 513, 578 or 600-608 us per launch depending on the physical frames a process happens to get (constant within a process); from one
// allocation the time is a deterministic function of the relative shift: 602 us at 0, 535-540 us at 32 .. 64 KiB per volume, 575-588 us
// at 128 / 224 KiB.  256^3 does not care (its working set lives in the Infinity Cache).  blub_fluid_desc::volume_shift_kib overrides.
template <class T>
static int vol_alloc(blub_fluid* h, T** p) {
    const size_t count = h->vol_cells;
    if (!h->slab) { int rc = dev_alloc_zero(h->stream, p, count); if (rc == BLUB_OK) { h->vol_owned.push_back(*p); *p -= h->vol_first; } return rc; }
    const size_t bytes = count * sizeof(T);
    size_t at = (h->slab_used + 0x1FFFFFull) & ~0x1FFFFFull;      // 2 MiB aligned ...
    at += ((size_t)h->slab_count * h->slab_shift) & 0x1FFFFFull;   // ... plus the shift of this volume
    if (at + bytes > h->slab_bytes) return set_error(BLUB_ERR_OUT_OF_MEMORY, "volume slab exhausted");
    *p = reinterpret_cast<T*>(h->slab + at) - h->vol_first;
    h->slab_used = at + bytes; h->slab_count += 1;
    HIP_TRY(hipMemsetAsync(h->slab + at, 0, bytes, h->stream));
    return BLUB_OK;
}
// the allocated planes of a volume, zeroed
template <class T>
static int vol_zero(blub_fluid* h, T* p) {
    HIP_TRY(hipMemsetAsync(p + h->vol_first, 0, h->vol_cells * sizeof(T), h->stream));
    return BLUB_OK;
}
static int solid_ensure(blub_fluid* h) {      // the solid-voxel volume exists only once a caller voxelises or uploads one
    if (h->solid) return BLUB_OK;
    HIP_TRY(hipMalloc((void**)&h->solid_alloc, h->vol_cells * sizeof(float4)));
    h->solid = h->solid_alloc - h->vol_first;
    return BLUB_OK;
}
static int copy_sync(blub_fluid* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BLUB_OK;
}

// ---- brick work lists ----------------------------------------------------------------------------------------------
// Asynchronous read-backs never use hipEvents: querying a pending event makes the runtime push marker packets into the
// stream, which costs milliseconds per step once the host runs ahead of the GPU.  Instead every snapshot carries the
// sequence number of the enqueue that produced it and the host simply polls the pinned (coherent) ring.

// phase: COMPACT_STEP_A (before P2G) / COMPACT_STEP_B (after advection) from the particle positions;
// COMPACT_ALL_ACTIVE (stand-alone stage calls): FLUID bricks from the marker volume, every brick active
static int build_lists(blub_fluid* h, int phase) {
    ProfScope ps(h, KC_BRICK_LISTS);
    // (brick_fluid is all zero here: allocated zeroed, and every build's scatter kernel clears it again)
    if (phase == COMPACT_ALL_ACTIVE)
        hipLaunchKernelGGL(k_bricks_mark_from_marker, dim3(h->bg.nb), dim3(BRICK_THREADS), 0, h->stream, h->bg, (const int8_t*)h->marker, h->brick_fluid);
    else if (h->bricks_premarked) h->bricks_premarked = false;     // k_advect marked them (stage_advect)
    else if (h->num_particles + h->num_ghost)
        hipLaunchKernelGGL(k_bricks_mark_particles, dim3(particle_blocks(h->num_particles + h->num_ghost)), dim3(256), 0, h->stream, h->bg, h->num_particles + h->num_ghost, (const float4*)h->pos, h->brick_fluid,
                           (const uint32_t*)h->n_dev, N_ALL);
    const int nblk = (h->bg.nb + 1023) / 1024;
    const int all_touched = (phase == COMPACT_ALL_ACTIVE) ? 1 : (int)h->all_touched;
    h->counts_seq += 1;
    if (nblk <= h->num_cus && !h->two_kernel_build) {      // every block co-resident: classification and scatter in ONE launch (k_bricks_build)
        hipLaunchKernelGGL(k_bricks_build, dim3(nblk), dim3(1024), 0, h->stream, h->bg, phase, all_touched, h->slab_z0 / BZ, h->slab_z1 / BZ, h->brick_fluid, h->brick_active,
                           h->brick_touched, reinterpret_cast<uint32_t*>(h->brick_block_counts), h->brick_block_ready, h->list_fluid, h->list_active, h->list_reset, h->counts,
                           h->counts_seq, h->counts_host_dev + (h->counts_seq % COUNTS_RING), &(h->counts_host_dev + COUNTS_RING)->pad0, h->resort_cursor);
        if (phase == COMPACT_STEP_A) h->all_touched = false;
        return BLUB_OK;
    }
    hipLaunchKernelGGL(k_bricks_classify, dim3(nblk), dim3(1024), 0, h->stream, h->bg, phase, all_touched, h->slab_z0 / BZ, h->slab_z1 / BZ, (const uint8_t*)h->brick_fluid, h->brick_active,
                       h->brick_touched, h->brick_flags, h->brick_block_counts);
    hipLaunchKernelGGL(k_bricks_scatter, dim3(nblk), dim3(1024), 0, h->stream, h->bg, (const uint8_t*)h->brick_flags, (const uint4*)h->brick_block_counts, nblk,
                       h->list_fluid, h->list_active, h->list_reset, h->counts, h->counts_seq, h->brick_fluid, h->counts_host_dev + (h->counts_seq % COUNTS_RING), h->resort_cursor);
    if (phase == COMPACT_STEP_A) h->all_touched = false;
    return BLUB_OK;
}
static int build_lists_from_particles(blub_fluid* h, int phase) { return build_lists(h, phase); }
// particles were changed from outside a step (or a stage ran on its own): marks a kernel left for the next list build are void
static int drop_brick_marks(blub_fluid* h) {
    if (!h->bricks_premarked) return BLUB_OK;
    h->bricks_premarked = false;
    HIP_TRY(hipMemsetAsync(h->brick_fluid, 0, (size_t)h->bg.nb, h->stream));
    return BLUB_OK;
}
static int build_lists_from_marker(blub_fluid* h) { return build_lists(h, COMPACT_ALL_ACTIVE); }
// newest snapshot of the brick counts that has landed (never waits unless `block`): only steers a performance choice
static int bricks_timeout_check(blub_fluid* h) {
    if (h->counts_host && ((const volatile BrickCounts*)h->counts_host)[COUNTS_RING].pad0) {
        h->counts_host[COUNTS_RING].pad0 = 0;
        h->num_cus = 0;   // (two-kernel build from now on)
        return set_error(BLUB_ERR_DEVICE, "a brick list build timed out waiting for its workgroups (device shared or partitioned?): the step that contained it is invalid; later steps use the two-kernel build");
    }
    return BLUB_OK;
}
static int latest_counts(blub_fluid* h, bool block, BrickCounts* out, bool* have) {
    *have = false;
    if (h->counts_seq == 0) return BLUB_OK;
    if (block) HIP_TRY(hipStreamSynchronize(h->stream));
    for (uint32_t k = 0; k < COUNTS_RING && k < h->counts_seq; ++k) {
        const uint32_t seq = h->counts_seq - k;
        const volatile BrickCounts* c = &h->counts_host[seq % COUNTS_RING];
        BrickCounts snap;
        snap.n_fluid = c->n_fluid; snap.n_active = c->n_active; snap.n_reset = c->n_reset; snap.n_stale = c->n_stale; snap.seq = c->seq; snap.seq_check = c->seq_check; snap.pad0 = c->pad0;
        if (snap.seq == seq && snap.seq_check == seq) {
            if (snap.pad0) {      // k_bricks_build gave up waiting for its own blocks: not co-resident on this device
                h->num_cus = 0;   // (two-kernel build from now on)
                return set_error(BLUB_ERR_DEVICE, "a brick list build timed out waiting for its workgroups (device shared or partitioned?): the step that contained it is invalid; later steps use the two-kernel build");
            }
            h->last_counts = snap; h->have_last_counts = true;
            *out = snap; *have = true; return BLUB_OK;
        }
    }
    // the host is a whole ring of list builds ahead of the device (short steps, several in flight): the last snapshot seen is still the best estimate
    if (h->have_last_counts) { *out = h->last_counts; *have = true; }
    return BLUB_OK;
}

#define LIST(h, which) (const uint32_t*)(h)->list_##which, (const uint32_t*)&(h)->counts->n_##which

// Launch grids of the kernels that loop over a brick list: the list lengths live on the device, so the grid is an ESTIMATE from the newest
// counts that have landed (+12 %), like pcg_brick_grid -- any grid is correct (grid-stride loops), it only costs speed.  One workgroup per
// brick matters: these kernels are one dependent load chain per brick (~4-8 us), and with the fixed grid of 2048 the 2600 active bricks of
// the headline scene made a quarter of the workgroups run two chains back to back (k_extrapolate_b: 16.7 us per launch against 8.7 us while
// the list still fitted).
struct ListGrids { int fluid, active, reset; };
static ListGrids list_grids(blub_fluid* h) {
    ListGrids lg{h->brick_grid, h->brick_grid, h->brick_grid};
    if (h->list_grid_forced > 0) { lg.fluid = lg.active = lg.reset = std::min(h->list_grid_forced, h->bg.nb); return lg; }
    BrickCounts bc; bool have = false;
    if (latest_counts(h, false, &bc, &have) != BLUB_OK || !have) return lg;      // (an error is reported by the solve stage, which asks too)
    auto sz = [&](uint32_t n) { return (int)std::min<uint32_t>(((uint32_t)h->bg.nb + 7u) & ~7u, (std::max<uint32_t>(256u, n + n / 8u + 16u) + 7u) & ~7u); };      // (a multiple of 8: list_slot's XCD-contiguous order)
    lg.fluid = sz(bc.n_fluid); lg.active = sz(bc.n_active); lg.reset = sz(bc.n_reset);
    return lg;
}

// ---- stages ------------------------------------------------------------------------------------------------------
static int stage_transfer(blub_fluid* h, float dt) {   // hybrid_fluid.rs:806-833
    int rc = build_lists_from_particles(h, COMPACT_STEP_A);
    if (rc != BLUB_OK) return rc;
    const ListGrids lg = list_grids(h);
    LAUNCH(h, KC_RESET_BRICKS, k_reset_bricks, dim3(lg.reset), dim3(BRICK_THREADS), h->bg, LIST(h, reset), (const float4*)h->solid, h->marker, h->ll[0], h->ll[1], h->ll[2],
           h->vel[0], h->vel[1], h->vel[2], h->pressure[0], h->pressure[1]);
    const uint32_t np_all = h->num_particles + h->num_ghost;
    if (np_all)
        LAUNCH(h, KC_BUILD_LISTS, k_build_lists, dim3(particle_blocks(np_all)), dim3(256), h->g, np_all, h->pos, h->marker,
               h->ll[0], h->ll[1], h->ll[2], (const float4*)h->pvel[0], (const float4*)h->pvel[1], (const float4*)h->pvel[2], h->nodes, h->node_stride, (int)(h->solid == nullptr), (int)h->standalone_stage, (const uint32_t*)h->n_dev, N_ALL);
    {
        GatherArgs3 a;
        for (int c = 0; c < 3; ++c) { a.heads[c] = h->ll[c]; a.out[c] = h->vel[c]; a.gravity_dt[c] = h->gravity[c] * dt; }
        a.node_stride = h->node_stride;
        const dim3 ggrid(3 * ((lg.active + 7) / 8) * 8);
        // Which of the two (bit-identical) gathers: the compacting one wins while a FLUID brick holds particles in a minority of its cells (headline scene: ~750
        // particles per FLUID brick, 69 against 76 us), one lane per list cell when the bricks are full (M4: ~4000 per brick, 3.7 against 4.1 ms).
        // Decided from the newest brick counts that have landed -- a speed choice only.
        bool compact = h->p2g_compact == 1;
        if (h->p2g_compact < 0) {
            BrickCounts bc; bool have = false;
            if ((rc = latest_counts(h, false, &bc, &have)) != BLUB_OK) return rc;
            compact = have && (uint64_t)np_all < 1536ull * bc.n_fluid;
        }
        // List-centric where the walk dominates (well-filled bricks: dam_halfhalf_highres 490 -> 436 us, M4 3.01 -> 2.60 ms incl. the finishing kernel); the
        // headline scene's sparse bricks are bound by per-tile latency, and a second launch costs more than the shorter walks save (66 -> 83 us): tile-centric there.
        const bool own = !compact && h->p2g_own != 0 && h->num_ghost == 0 && h->n_dev == nullptr && h->slab_z0 <= 0 && h->slab_z1 >= h->g.nz;      // (slabs: the lists of the brick layer below belong to nobody's work list)
        if (own) {
            if (!h->gather_sums) {
                if ((rc = dev_alloc_zero(h->stream, &h->gather_sums, (size_t)3 * h->bg.nb * GT_N)) != BLUB_OK || (rc = dev_alloc_zero(h->stream, &h->gather_stamp, (size_t)3 * h->bg.nb)) != BLUB_OK) return rc;
            }
            GatherOwnArgs3 o;
            for (int c = 0; c < 3; ++c) { o.heads[c] = h->ll[c]; o.out[c] = h->vel[c]; o.gravity_dt[c] = a.gravity_dt[c]; }
            o.node_stride = h->node_stride;
            o.halo.sums = h->gather_sums; o.halo.stamp = h->gather_stamp; o.halo.seq = h->counts_seq; o.halo.nb = (uint32_t)h->bg.nb;
            LAUNCH(h, KC_GATHER_VELOCITY, k_gather_velocity3_po, ggrid, dim3(BX * BY * BZ), h->bg, LIST(h, active), (const int8_t*)h->marker, (const blubk::GatherNode*)h->nodes, o);
            LAUNCH(h, KC_GATHER_VELOCITY, k_gather_finish3, ggrid, dim3(256), h->bg, LIST(h, active), (const int8_t*)h->marker, o);
        }
        else if (compact) LAUNCH(h, KC_GATHER_VELOCITY, k_gather_velocity3_s, ggrid, dim3(GS_THREADS), h->bg, LIST(h, active), (const int8_t*)h->marker, (const blubk::GatherNode*)h->nodes, a);
        else LAUNCH(h, KC_GATHER_VELOCITY, k_gather_velocity3_p, ggrid, dim3(768), h->bg, LIST(h, active), (const int8_t*)h->marker, (const blubk::GatherNode*)h->nodes, a);
    }
    return BLUB_OK;
}
static DivergenceSrc divergence_src(const blub_fluid* h) { return DivergenceSrc{h->vel[0], h->vel[1], h->vel[2], (const float4*)h->solid}; }
// `defer`: inside blub_fluid_step the divergence is not launched here -- the velocity solve that follows forms it inside its init kernel when it runs on
// the brick mapping (k_pcg_init_b<true>: one launch and one pass over the residual volume less) and launches this kernel itself otherwise
static int stage_divergence(blub_fluid* h, bool defer = false) {   // :836-840
    if (defer && h->fuse_divergence) { h->divergence_deferred = true; return BLUB_OK; }
    h->divergence_deferred = false;
    LAUNCH(h, KC_DIVERGENCE, k_divergence_b, dim3(list_grids(h).fluid), dim3(BRICK_THREADS), h->bg, LIST(h, fluid), (const int8_t*)h->marker, divergence_src(h), h->residual);
    return BLUB_OK;
}

// `written_by_kernel`: k_pcg_finalize already stored the sample into the pinned ring slot of this solve
static int enqueue_stats_readback(blub_fluid* h, int which, float dt, bool written_by_kernel = false) {   // enqueue_error_buffer_read, pressure_solver.rs:176-191
    if ((int)h->stats_pending[which].size() < STATS_RING) {
        const uint32_t seq = h->solve_seq[which];
        const int slot = (int)(seq % STATS_RING);
        if (!written_by_kernel) HIP_TRY(hipMemcpyAsync(&h->stats_host[which][slot], h->ctrl[which], sizeof(PcgCtrl), hipMemcpyDeviceToHost, h->stream));
        h->stats_pending[which].push_back({seq, slot});
        h->stats_dt[which].push_back(dt);
    }   // else: "No more error buffer available" -- the reference warns and skips the sample (:188-190)
    return BLUB_OK;
}

// PressureSolver::solve for the LOD0 preconditioner reading: the literal kernel sequence (dense rows, marker based).
static int stage_solve_lod0(blub_fluid* h, int which, float dt) {
    float* p = h->pressure[which];
    const blub_solver_config& c = h->cfg[which];
    const float tol = c.error_tolerance / dt;
    PcgCtrl* ctrl = h->ctrl[which];
    const dim3 grid(h->pcg_grid), block(256);
    const int np = h->pcg_grid;
    LAUNCH(h, KC_PCG_LOD0, k_pcg_init, grid, block, h->geom, h->marker, p, h->residual, h->search, (float*)nullptr, h->tile_flags);
    LAUNCH(h, KC_PCG_LOD0, k_pcg_precond_lod0, grid, block, h->geom, h->marker, h->residual, h->aux_temp, (const float*)nullptr, (float*)nullptr, h->tile_flags, ctrl);
    LAUNCH(h, KC_PCG_LOD0, k_pcg_precond_lod0, grid, block, h->geom, h->marker, h->aux_temp, h->search, (const float*)h->residual, h->part_sigma[0], h->tile_flags, ctrl);
    const int maxit = c.max_num_iterations;
    for (int i = 0; i <= maxit; ++i) {   // :654-723
        float* sig_cur = h->part_sigma[i & 1];
        float* sig_next = h->part_sigma[(i + 1) & 1];
        LAUNCH(h, KC_PCG_LOD0, k_pcg_apply, grid, block, h->geom, h->marker, h->search, h->part_sas, h->tile_flags, ctrl);
        const int last = (i == maxit);
        const int check = last || (i > 0 && c.error_check_frequency > 0 && i % c.error_check_frequency == 0);   // :672-673
        LAUNCH(h, KC_PCG_LOD0, k_pcg_update, grid, block, h->geom, h->marker, h->search, p, h->residual, h->part_sas, sig_cur, sig_next, h->part_max, np, h->tile_flags, ctrl);
        if (!last) {
            LAUNCH(h, KC_PCG_LOD0, k_pcg_precond_lod0, grid, block, h->geom, h->marker, h->residual, h->aux_temp, (const float*)nullptr, (float*)nullptr, h->tile_flags, ctrl);
            LAUNCH(h, KC_PCG_LOD0, k_pcg_precond_lod0, grid, block, h->geom, h->marker, h->aux_temp, h->aux, (const float*)h->residual, sig_next, h->tile_flags, ctrl);
        }
        LAUNCH(h, KC_PCG_LOD0, k_pcg_search, grid, block, h->geom, h->marker, h->aux, h->search, sig_cur, sig_next, h->part_max, np, h->tile_flags, ctrl, tol, i, check, last);
        if (last) break;
    }
    LAUNCH(h, KC_PCG_LOD0, k_pcg_tag, dim3(1), dim3(1), ctrl, h->solve_seq[which]);
    return BLUB_OK;
}

static int ensure_pcg1_buffers(blub_fluid* h) {
    int rc = BLUB_OK;
    for (int k = 0; k < 3 && rc == BLUB_OK; ++k) if (!h->cgbuf_alloc[k]) {
        if (h->mem_mode == BLUB_SLAB_MEMORY_COARSE) rc = dev_alloc_zero(h->stream, &h->cgbuf_alloc[k], h->vol_cells);
        else {
            if (shared_malloc((void**)&h->cgbuf_alloc[k], h->vol_cells * sizeof(float), h->mem_mode) != hipSuccess) { h->cgbuf_alloc[k] = nullptr; (void)hipGetLastError(); rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "allocation of a recurrence volume failed"); }
            else HIP_TRY(hipMemsetAsync(h->cgbuf_alloc[k], 0, h->vol_cells * sizeof(float), h->stream));
        }
        if (rc == BLUB_OK) h->cgbuf[k] = h->cgbuf_alloc[k] - h->vol_first;
    }
    if (rc == BLUB_OK && !h->part4) rc = dev_alloc_zero(h->stream, &h->part4, 2 * (size_t)PCG_GRID_MAX);
    for (int w = 0; w < 2 && rc == BLUB_OK; ++w) if (!h->pcg1_scalars[w]) rc = dev_alloc_zero(h->stream, &h->pcg1_scalars[w], 1);
    return rc;
}

// Launch grid of the brick-mapped PCG kernels: an ESTIMATE of the number of virtual workgroups (pcg_vblocks) from the newest landed FLUID brick
// count, with 12 % head room.  It only costs speed when it is off: surplus workgroups exit, missing ones are covered by the others' loops, and
// the dot-product grouping is fixed by the device-side list alone (blub_pcg.hip.h).
static int pcg_brick_grid(const blub_fluid* h, bool have, const BrickCounts& bc) {
    if (h->pcg_grid_forced > 0) return (std::min(h->pcg_grid_forced, PCG_GRID_MAX) + 7) & ~7;
    int np = std::min((h->bg.nb + PCG_BPB - 1) / PCG_BPB, PCG_VBLOCKS_MAX);
    if (have) np = std::max(64, std::min(np, (int)((bc.n_fluid * 9u / 8u + 8u + (unsigned)PCG_BPB - 1u) / (unsigned)PCG_BPB)));
    return (np + 7) & ~7;   // (a multiple of 8: the XCD-contiguous list order of the single-reduction kernels)
}

// PressureSolver::solve, pressure_solver.rs:591-729 (schedule: SURVEY Appendix D), fused iteration kernels (blub_pcg*.hip.h)
static int stage_solve(blub_fluid* h, int which, float dt, bool standalone) {
    float* p = h->pressure[which];
    const blub_solver_config& c = h->cfg[which];
    if (!h->pressure_initialised[which]) {   // :601-603
        { int rz = vol_zero(h, p); if (rz != BLUB_OK) return rz; }
        h->pressure_initialised[which] = true;
    }
    const float tol = c.error_tolerance / dt;   // :197
    PcgCtrl* ctrl = h->ctrl[which];
    h->solve_seq[which] += 1;
    int rc;
    const bool div_pending = which == 0 && h->divergence_deferred;
    if (h->precond_mode != BLUB_PRECOND_ZERO) {
        if (div_pending && (rc = stage_divergence(h)) != BLUB_OK) return rc;
        HIP_TRY(hipMemsetAsync(ctrl, 0, sizeof(PcgCtrl), h->stream));
        h->last_schedule[which] = 0; h->last_mapping[which] = 2;
        if ((rc = stage_solve_lod0(h, which, dt)) != BLUB_OK) return rc;
        return enqueue_stats_readback(h, which, dt);
    }
    // work mapping: brick lists when the fluid is sparse, dense rows otherwise (a performance choice only)
    bool sparse;
    BrickCounts bc{}; bool have = false;
    if ((rc = latest_counts(h, standalone || h->counts_seq <= 2, &bc, &have)) != BLUB_OK) return rc;
    if (h->force_pcg_path >= 0) sparse = h->force_pcg_path >= 1;
    else {
        // Measured per iteration on compact all-FLUID slabs (tools/mapping_crossover.py, profiles/r03_mapping_crossover.txt): the brick mapping costs
        // ~6 us per 1000 FLUID bricks (+12-18 us), the dense one follows the cells of the grid; with the single-reduction schedule (ONE kernel per
        // iteration) the bricks win up to ~8000 FLUID bricks whatever the grid (256x128x128: 23 vs 27 us half full, 45 vs 40 us full; dam_halfhalf_highres:
        // 344 vs 270 steps/s) and up to ~30 % of the bricks on large grids; with the reference's two-reduction order only up to ~20 % / ~2000 bricks
        // (256^3 a quarter full: 62 vs 53 us).  Grids of up to 1 M cells stay launch/latency-bound however full they are: always bricks there
        // (dam_halfhalf, 128x64x64, 40-75 % of the bricks FLUID while it sloshes: 1647 steps/s on the brick mapping, 1115 on the dense one).
        const bool single = h->pcg_schedule == 1 && c.max_num_iterations <= h->pcg1_max_iterations;
        const float limit = std::max(single ? 8192.0f : 2048.0f, (single ? SPARSE_PCG_MAX_FILL : SPARSE_PCG_MAX_FILL_REFERENCE) * (float)h->bg.nb);
        sparse = (have && (float)bc.n_fluid < limit) || h->N <= (size_t)1 << 20 || h->gz.tiles < 256;   // (tiny grids: too few dense tiles to fill the chip)
    }
    if (2 * h->gz.qpr > h->gz.T) sparse = true;   // rows wider than 2048 cells: the dense tiles cannot hold their halo rows (k_pcg_dir_z)
    const int maxit = c.max_num_iterations;
    const int freq = c.error_check_frequency;
    auto is_check = [&](int j) { return j > 0 && freq > 0 && j % freq == 0; };   // :672-673 (the i == max case is k_pcg_finalize)
    float* sbuf[2] = {h->search, h->aux};
    // the statistics sample of this solve goes straight into its pinned ring slot (if the ring has room)
    const bool ring_free = (int)h->stats_pending[which].size() < STATS_RING;
    PcgCtrl* stat_slot = ring_free ? h->stats_host_dev[which] + (h->solve_seq[which] % STATS_RING) : (PcgCtrl*)nullptr;
    float2* part_upd = reinterpret_cast<float2*>(h->part_sigma[0]);   // {(M^-1 r).r, max|r|} partials (init / update kernels)
    float* part_dir = h->part_sas;                                    // s.As partials (direction kernel)
    if (div_pending && !sparse && (rc = stage_divergence(h)) != BLUB_OK) return rc;     // the dense mapping reads b from the residual volume
    h->last_mapping[which] = sparse ? 1 : 0;
    h->last_schedule[which] = (sparse && h->pcg_schedule == 1 && c.max_num_iterations <= h->pcg1_max_iterations) ? 1 : 0;
    if (sparse) {
        const int np = pcg_brick_grid(h, have, bc);
        const dim3 grid(np), block(PCG_B_THREADS);
        const uint32_t* nfl = &h->counts->n_fluid;
        if (div_pending) {
            h->divergence_deferred = false;
            LAUNCH(h, KC_PCG_INIT, k_pcg_init_b<true>, grid, block, h->bg, LIST(h, active), nfl, 0, (const int8_t*)h->marker, h->dvol, p, h->residual, sbuf[0], part_upd, ctrl, h->tail_sync[which], divergence_src(h));
        } else
            LAUNCH(h, KC_PCG_INIT, k_pcg_init_b<false>, grid, block, h->bg, LIST(h, active), nfl, 0, (const int8_t*)h->marker, h->dvol, p, h->residual, sbuf[0], part_upd, ctrl, h->tail_sync[which], DivergenceSrc{});
        if (h->pcg_schedule == 1 && maxit <= h->pcg1_max_iterations) {
            // ONE kernel per iteration (blub_pcg1.hip.h): r, w = A M^-1 r and q = A d are double buffered by iteration parity,
            // the search direction d (BLUB_VOLUME_SEARCH) and p are updated in place
            if ((rc = ensure_pcg1_buffers(h)) != BLUB_OK) return rc;
            float* R[2] = {h->residual, h->cgbuf[0]};
            float* W[2] = {h->aux, h->cgbuf[1]};
            float* Q[2] = {h->aux_temp, h->cgbuf[2]};
            float4* part[2] = {h->part4, h->part4 + PCG_GRID_MAX};
            Pcg1Scalars* sc = h->pcg1_scalars[which];
            SlabDirect nodir{}; nodir.log = h->scalar_log[which]; nodir.stamps = h->phase_stamps[which];
            if (nodir.stamps) HIP_TRY(hipMemsetAsync(nodir.stamps, 0, 64 * 8 * sizeof(unsigned long long), h->stream));
            if (nodir.log) HIP_TRY(hipMemsetAsync(nodir.log, 0xFF, 1024 * sizeof(float4), h->stream));
            LAUNCH(h, KC_PCG_INIT, k_pcg1_w0_s<false>, grid, block, h->bg, LIST(h, fluid), 0, (const uint8_t*)h->dvol, (const float*)h->search, W[0], (const float2*)part_upd, 0, part[0], 1, 0u, -1, -1, SlabDirect{});
            // Launch as many iterations as the last few solves needed (+ `tail_margin_checks` check intervals); ONE persistent kernel covers
            // the rest (k_pcg1_tail_s): it normally finds the solve finished and only publishes the statistics.  Only while the solve is
            // launch-bound (an iteration inside the tail -- <= 256 blocks, a grid barrier -- costs more than a launched one).
            int launched1 = maxit + 1;
            if (h->use_tail && (!have || bc.n_fluid <= 2048u) && freq > 0 && !h->stats_history[which].empty()) {
                int recent = 0, k = 0;
                for (auto it2 = h->stats_history[which].rbegin(); it2 != h->stats_history[which].rend() && k < 4; ++it2, ++k) recent = std::max(recent, (int)it2->iteration_count);
                if (recent >= 0 && recent < maxit) launched1 = std::min(maxit + 1, (recent / freq + h->tail_margin_checks) * freq + 2);   // K(c + 1) forms the verdict of check c
            }
            if (h->use_tail && h->tail_first_forced >= 0) launched1 = std::min(maxit + 1, std::max(1, h->tail_first_forced));   // (test hook; K(0) is always launched)
            for (int i = 0; i < launched1; ++i) {
                const float4* pin = part[i & 1]; float4* pout = part[(i + 1) & 1];
                if (i == 0) LAUNCH(h, KC_PCG_ITER, (k_pcg1_iter_s<true>), grid, block, h->bg, LIST(h, fluid), 0, (const uint8_t*)h->dvol, (const float*)R[0], R[1], (const float*)W[0], W[1], (const float*)Q[1], Q[0], h->search, p, pin, pout, 0, ctrl, sc, tol, 0, 0, -1, -1, nodir);
                else LAUNCH(h, KC_PCG_ITER, (k_pcg1_iter_s<false>), grid, block, h->bg, LIST(h, fluid), 0, (const uint8_t*)h->dvol, (const float*)R[i & 1], R[(i + 1) & 1], (const float*)W[i & 1], W[(i + 1) & 1], (const float*)Q[(i + 1) & 1], Q[i & 1], h->search, p, pin, pout, 0, ctrl, sc, tol, i, (int)is_check(i - 1), -1, -1, nodir);
            }
            if (launched1 <= maxit) {
                if (h->tail_inject_timeout) { const int one = 1; HIP_TRY(hipMemcpyAsync(&h->tail_sync[which]->timed_out, &one, sizeof one, hipMemcpyHostToDevice, h->stream)); h->tail_inject_timeout = false; }
                const dim3 tgrid((unsigned)std::min(np, h->tail_grid));     // (tail_grid: a multiple of 8, every block co-resident)
                LAUNCH(h, KC_PCG_FINALIZE, k_pcg1_tail_s<true>, tgrid, block, h->bg, LIST(h, fluid), (const uint8_t*)h->dvol, R[0], R[1], W[0], W[1], Q[0], Q[1], h->search, p,
                       part[0], part[1], ctrl, sc, tol, launched1, maxit, freq, h->tail_sync[which], h->solve_seq[which], stat_slot);
            } else
                LAUNCH(h, KC_PCG_FINALIZE, k_pcg1_finalize, dim3(1), dim3(256), ctrl, (const float4*)part[(maxit + 1) & 1], 0, nfl, maxit, h->solve_seq[which], stat_slot, 0u, (uint32_t*)nullptr);
            // the residual of a full-length solve ends in R[(maxit + 1) & 1]; keep BLUB_VOLUME_RESIDUAL pointing at it
            if ((maxit + 1) & 1) std::swap(h->residual, h->cgbuf[0]);
            return enqueue_stats_readback(h, which, dt, true);
        }
        // the reference's two-reduction schedule: two kernels per iteration.  As for the single-reduction schedule the host launches the pairs the
        // last few solves needed (+ `tail_margin_checks` check intervals) and ONE persistent kernel for the rest (k_pcg_tail_s): it normally finds
        // the solve finished and only publishes the statistics; without it a finished solve still cost 2 x (max - needed) no-op launches.
        int launched = maxit + 1;
        if (h->use_tail && (!have || bc.n_fluid <= 2048u) && freq > 0 && !h->stats_history[which].empty()) {
            int recent = 0, k = 0;
            for (auto it2 = h->stats_history[which].rbegin(); it2 != h->stats_history[which].rend() && k < 4; ++it2, ++k) recent = std::max(recent, (int)it2->iteration_count);
            if (recent >= 0 && recent < maxit) launched = std::min(maxit + 1, (recent / freq + h->tail_margin_checks) * freq + 1);   // KD(c + 1) forms the verdict of check c
        }
        if (h->use_tail && h->tail_first_forced >= 0) launched = std::min(maxit + 1, std::max(1, h->tail_first_forced));   // (test hook; iteration 0 is always launched)
        for (int i = 0; i < launched; ++i) {
            if (i == 0)
                LAUNCH(h, KC_PCG_DIR, k_pcg_dir_s<true>, grid, block, h->bg, LIST(h, fluid), 0, (const uint8_t*)h->dvol, (const float*)h->residual, (const float*)sbuf[0], sbuf[0],
                       (const float2*)part_upd, part_dir, 0, ctrl, tol, i, 0, -1, -1);
            else
                LAUNCH(h, KC_PCG_DIR, k_pcg_dir_s<false>, grid, block, h->bg, LIST(h, fluid), 0, (const uint8_t*)h->dvol, (const float*)h->residual, (const float*)sbuf[(i - 1) & 1], sbuf[i & 1],
                       (const float2*)part_upd, part_dir, 0, ctrl, tol, i, (int)is_check(i - 1), -1, -1);
            LAUNCH(h, KC_PCG_UPDATE, k_pcg_update_s, grid, block, h->bg, LIST(h, fluid), 0, (const uint8_t*)h->dvol, (const float*)sbuf[i & 1], p, h->residual,
                   (const float*)part_dir, part_upd, 0, (const PcgCtrl*)ctrl, i);
        }
        if (launched <= maxit) {
            if (h->tail_inject_timeout) { const int one = 1; HIP_TRY(hipMemcpyAsync(&h->tail_sync[which]->timed_out, &one, sizeof one, hipMemcpyHostToDevice, h->stream)); h->tail_inject_timeout = false; }
            const dim3 tgrid((unsigned)std::min(np, h->tail_grid));
            LAUNCH(h, KC_PCG_FINALIZE, k_pcg_tail_s, tgrid, block, h->bg, LIST(h, fluid), (const uint8_t*)h->dvol, h->residual, sbuf[0], sbuf[1], p, part_upd, part_dir, ctrl, tol,
                   launched, maxit, freq, h->tail_sync[which], h->solve_seq[which], stat_slot);
        } else
            LAUNCH(h, KC_PCG_FINALIZE, k_pcg_finalize, dim3(1), dim3(256), ctrl, (const float2*)part_upd, 0, nfl, maxit, h->solve_seq[which], stat_slot);
    } else {
        const int np = h->pcg_grid_z;
        const int npd = std::min(np, ((h->gzd.tiles + 7) / 8) * 8);      // the direction kernel's own grid (it may march deeper tiles: set_dense_geometry)
        const dim3 grid(np), gridd(npd);
        const size_t lds_dir = dense_dir_lds_bytes(h->gz.T, h->gz.qpr);
#define BLUB_LAUNCH_Z(TT, NTU, NTD, DD)                                                                                                                           \
        {                                                                                                                                                       \
            const dim3 block(TT);                                                                                                                               \
            LAUNCH(h, KC_PCG_INIT, k_pcg_init_z<TT>, grid, block, h->gz, (const int8_t*)h->marker, h->dvol, p, h->residual, sbuf[0], part_upd, h->tile_flags, ctrl);   \
            for (int i = 0; i <= maxit; ++i) {                                                                                                                  \
                if (i == 0)                                                                                                                                     \
                    LAUNCH_LDS(h, KC_PCG_DIR, (k_pcg_dir_z<TT, true, NTD, DD>), gridd, block, lds_dir, h->gzd, (const uint8_t*)h->dvol, (const float*)h->residual, (const float*)sbuf[0], sbuf[0],   \
                           (const float2*)part_upd, part_dir, np, (const uint8_t*)h->tile_flags, ctrl, tol, i, 0);                                              \
                else                                                                                                                                            \
                    LAUNCH_LDS(h, KC_PCG_DIR, (k_pcg_dir_z<TT, false, NTD, DD>), gridd, block, lds_dir, h->gzd, (const uint8_t*)h->dvol, (const float*)h->residual, (const float*)sbuf[(i - 1) & 1], sbuf[i & 1], \
                           (const float2*)part_upd, part_dir, np, (const uint8_t*)h->tile_flags, ctrl, tol, i, (int)is_check(i - 1));                           \
                LAUNCH(h, KC_PCG_UPDATE, (k_pcg_update_z<TT, NTU>), grid, block, h->gz, (const uint8_t*)h->dvol, (const float*)sbuf[i & 1], p, h->residual,     \
                       (const float*)part_dir, part_upd, npd, (const uint8_t*)h->tile_flags, (const PcgCtrl*)ctrl, i);                                           \
            }                                                                                                                                                   \
        }
        // p / r of KU are touched exactly once per kernel: non-temporal (66.8 -> 62.5 us at 256^3); s_out of KD is re-read as a halo: default policy
        // s_out of KD: default cache policy while the iteration's working set fits the 256 MiB Infinity Cache (KU re-reads it: 256^3 KU 55.4 vs 58.0 us),
        // non-temporal beyond (512^3: KD 316 vs 324 us, KU 524 vs 538 us)
        const bool kd_nt = h->dense_kd_nt > 0 || (h->dense_kd_nt < 0 && h->N >= ((size_t)1 << 26));
        // (raw planes in flight per quad of the direction kernel: 2.  Measured with 3 and 4 -- 139 / 159 VGPRs, one wave of occupancy less each --:
        //  46.1 / 46.6 us against 39.0 us at 256^3, 345 against 316 us at 512^3, also with 32- and 64-plane tiles)
        if (h->gz.T == 256) { if (kd_nt) BLUB_LAUNCH_Z(256, true, true, 2) else BLUB_LAUNCH_Z(256, true, false, 2) }
        else if (h->gz.T == 1024) { if (kd_nt) BLUB_LAUNCH_Z(1024, true, true, 2) else BLUB_LAUNCH_Z(1024, true, false, 2) }
        else { if (kd_nt) BLUB_LAUNCH_Z(512, true, true, 2) else BLUB_LAUNCH_Z(512, true, false, 2) }
#undef BLUB_LAUNCH_Z
        LAUNCH(h, KC_PCG_FINALIZE, k_pcg_finalize, dim3(1), dim3(256), ctrl, (const float2*)part_upd, np, (const uint32_t*)nullptr, maxit, h->solve_seq[which], stat_slot);
    }
    // the search direction of a full-length solve ends in sbuf[maxit & 1]; keep BLUB_VOLUME_SEARCH pointing at it
    if (maxit & 1) std::swap(h->search, h->aux);
    return enqueue_stats_readback(h, which, dt, true);
}

static int stage_binning(blub_fluid* h) {   // hybrid_fluid.rs:857-893
    if (h->binning_mode == BLUB_BINNING_OFF || h->num_particles == 0) return BLUB_OK;
    // The reference counts and scans in its linked-list volume (:858-876) because the dense transfer_clear pass re-zeroes it
    // anyway.  Here list heads are only re-zeroed inside the reset-list bricks, so the per-cell counters / prefix sums live in a
    // volume that is scratch at this point of the step instead (aux_temp: only ever read inside a solve, after being rewritten) and
    // ll[0] keeps the invariant "zero outside the touched bricks".
    // BLUB_BINNING_LITERAL (Q4): the dispatches cover ceil(P / 64) * 64 threads without a guard -- the records behind the live range
    // (zeros, or what an earlier pass left there) are binned too --, destinations are 1-based, and the WHOLE buffer is copied back.
    const bool literal = h->binning_mode == BLUB_BINNING_LITERAL;
    const uint32_t T = literal ? std::min<uint32_t>((h->num_particles + 63u) / 64u * 64u, h->max_particles) : h->num_particles;
    uint32_t* counters = reinterpret_cast<uint32_t*>(h->aux_temp);
    { int rz = vol_zero(h, counters); if (rz != BLUB_OK) return rz; }   // clear_texture :858
    LAUNCH(h, KC_BIN_COUNT, k_bin_count, dim3(particle_blocks(T)), dim3(256), h->g, T, h->pos, counters, (const uint32_t*)h->n_dev, N_OWN);
    const int n = (int)h->vol_cells, nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;      // (a slab scans its own planes: all its particles live there)
    LAUNCH(h, KC_BIN_SCAN, k_scan_block_totals, dim3(nblocks), dim3(1024), (const uint32_t*)(counters + h->vol_first), n, h->scan_totals);
    LAUNCH(h, KC_BIN_SCAN, k_scan_totals, dim3(1), dim3(1024), h->scan_totals, nblocks);
    LAUNCH(h, KC_BIN_SCAN, k_scan_apply, dim3(nblocks), dim3(1024), counters + h->vol_first, n, (const uint32_t*)h->scan_totals);
    LAUNCH(h, KC_BIN_REWRITE, k_bin_rewrite, dim3(particle_blocks(T)), dim3(256), h->g, T, h->max_particles,
           (const float4*)h->pos, h->pos_tmp, (const uint32_t*)counters, (int)literal, (const uint32_t*)h->n_dev, N_OWN);
    {
        ProfScope ps(h, KC_COPY);   // :885-891 ("fixed": only the live range; the rest of the buffer is never read)
        HIP_TRY(hipMemcpyAsync(h->pos, h->pos_tmp, (size_t)(literal ? h->max_particles : h->num_particles) * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
    }
    return BLUB_OK;
}
// ---- internal particle order ---------------------------------------------------------------------------------------------------------------
static int ensure_resort_tables(blub_fluid* h) {
    if (h->pid) return BLUB_OK;
    const size_t P = std::max<size_t>(h->max_particles, 1), T = (size_t)h->bg.nb * (BX * BY * BZ) + 4;
    int rc = dev_alloc_zero(h->stream, &h->pid, P);
    if (rc == BLUB_OK) rc = dev_alloc_zero(h->stream, &h->pid_tmp, P);
    if (rc == BLUB_OK) rc = dev_alloc_zero(h->stream, &h->resort_counters, T);
    if (rc == BLUB_OK) rc = dev_alloc_zero(h->stream, &h->resort_starts, T);
    if (rc == BLUB_OK) rc = dev_alloc_zero(h->stream, &h->resort_ranks, P);
    if (rc != BLUB_OK) return rc;
    hipLaunchKernelGGL(k_iota, dim3(particle_blocks((uint32_t)P)), dim3(256), 0, h->stream, 0u, (uint32_t)P, h->pid);
    return BLUB_OK;
}
// the engine may re-sort on its own only where nobody else addresses particles by index: a single domain (slabs migrate particles between processes by
// slot), and not with the literal rebinning quirk (Q4 reads the records BEHIND the live range, whose contents depend on the history of the buffer)
static bool resort_allowed(const blub_fluid* h) {
    return h->resort_every > 0 && h->n_dev == nullptr && h->num_ghost == 0 && h->binning_mode != BLUB_BINNING_LITERAL && h->num_particles > 1 && !h->standalone_stage;
}
static int stage_resort(blub_fluid* h) {      // the fluid brick list of this step's transfer is current here: no particle has moved since
    int rc = ensure_resort_tables(h);
    if (rc != BLUB_OK) return rc;
    const uint32_t n = h->num_particles;
    LAUNCH(h, KC_BIN_COUNT, k_resort_count, dim3(particle_blocks(n)), dim3(256), h->bg, n, (const float4*)h->pos, h->resort_counters, h->resort_ranks);
    LAUNCH(h, KC_BIN_SCAN, k_resort_scan, dim3(list_grids(h).fluid / RESORT_GROUP + 1), dim3(BRICK_THREADS), h->bg, LIST(h, fluid), h->resort_counters, h->resort_starts, h->resort_cursor);      // (the cursor was zeroed by the list build of this step's transfer)
    LAUNCH(h, KC_BIN_REWRITE, k_resort_move, dim3(particle_blocks(n)), dim3(256), h->bg, n, (const float4*)h->pos, (const uint32_t*)h->pid, (const uint32_t*)h->resort_ranks,
           (const uint32_t*)h->resort_starts, h->pos_tmp, h->pid_tmp);
    std::swap(h->pos, h->pos_tmp); std::swap(h->pid, h->pid_tmp);
    h->order_internal = true; h->resorts_done += 1;
    return BLUB_OK;
}
// Every entry point that shows or takes particles BY INDEX calls this first: the arrays go back to the caller's order (and the indices stored in the
// list links / heads with them).  A no-op unless the engine has re-sorted since the order was last the caller's.
static int restore_canonical_order(blub_fluid* h) {
    if (!h->order_internal) return BLUB_OK;
    h->order_internal = false;
    const uint32_t n = h->num_particles;
    if (n) {
        ProfScope ps(h, KC_COPY);
        float4* arr[4] = {h->pos, h->pvel[0], h->pvel[1], h->pvel[2]};
        for (int a = 0; a < 4; ++a) {
            hipLaunchKernelGGL(k_unpermute, dim3(particle_blocks(n)), dim3(256), 0, h->stream, n, (const uint32_t*)h->pid, (const float4*)arr[a], h->pos_tmp, (int)(a == 0));
            HIP_TRY(hipMemcpyAsync(arr[a], h->pos_tmp, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        }
        hipLaunchKernelGGL(k_translate_heads, dim3(stream_blocks(h->vol_cells)), dim3(256), 0, h->stream, h->vol_cells, n, (const uint32_t*)h->pid, h->ll[0] + h->vol_first);
    }
    hipLaunchKernelGGL(k_iota, dim3(particle_blocks(h->max_particles)), dim3(256), 0, h->stream, 0u, h->max_particles, h->pid);
    return BLUB_OK;
}
static int stage_extrapolate(blub_fluid* h) {
    LAUNCH(h, KC_EXTRAPOLATE, k_extrapolate_b, dim3(list_grids(h).active), dim3(BRICK_THREADS), h->bg, LIST(h, active), (const int8_t*)h->marker, h->vel[0], h->vel[1], h->vel[2]);
    return BLUB_OK;
}
static int stage_project(blub_fluid* h) {   // :906-914
    LAUNCH(h, KC_DIVERGENCE_REMOVE, k_divergence_remove_b, dim3(list_grids(h).active), dim3(BRICK_THREADS), h->bg, LIST(h, active), (const int8_t*)h->marker,
           (const float*)h->pressure[0], (const float4*)h->solid, h->vel[0], h->vel[1], h->vel[2]);
    return stage_extrapolate(h);
}
static int stage_advect_particles(blub_fluid* h, float dt, bool insert_lists, bool mark_bricks = false) {   // :916-926
    // mark_bricks: the list build that follows takes its FLUID bricks from this kernel's marks (brick_fluid is all zero here, see build_lists)
    // the reset list (active + stale bricks of this step, own AND ghost bricks) is a superset of the active list
    LAUNCH(h, KC_RESET_BRICKS, k_reset_bricks, dim3(list_grids(h).reset), dim3(BRICK_THREADS), h->bg, LIST(h, reset), (const float4*)h->solid, h->marker, h->ll[0],
           (uint32_t*)nullptr, (uint32_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    if (h->num_particles) {
#define BLUB_ADVECT(F) LAUNCH(h, KC_ADVECT, k_advect<F>, dim3(particle_blocks(h->num_particles)), dim3(256), h->g, h->num_particles, dt, h->pos, h->pvel[0], h->pvel[1], h->pvel[2], \
               h->vel[0], h->vel[1], h->vel[2], h->solid, h->marker, insert_lists ? h->ll[0] : (uint32_t*)nullptr, \
               mark_bricks ? h->brick_fluid : (uint8_t*)nullptr, h->bg.nbx, h->bg.nby, (const uint32_t*)h->n_dev, N_OWN)
        if (h->filter_mode == BLUB_FILTER_WEIGHTED) BLUB_ADVECT(1); else if (h->filter_mode == BLUB_FILTER_WEIGHTED8) BLUB_ADVECT(2); else BLUB_ADVECT(0);
#undef BLUB_ADVECT
    }
    h->bricks_premarked = mark_bricks && h->num_particles != 0;
    return BLUB_OK;
}
static int stage_advect(blub_fluid* h, float dt) {   // :916-932
    int rc = stage_advect_particles(h, dt, true, h->num_ghost == 0);
    return rc != BLUB_OK ? rc : build_lists_from_particles(h, COMPACT_STEP_B);
}
static int stage_density_gather(blub_fluid* h, float dt) {   // :933-937
    LAUNCH(h, KC_DENSITY_GATHER, k_density_gather_p, dim3(list_grids(h).fluid), dim3(768), h->bg, LIST(h, fluid), (const int8_t*)h->marker, (const uint32_t*)h->ll[0],
           (const float4*)h->pos, h->residual, dt);
    return BLUB_OK;
}
static int stage_position_change(blub_fluid* h, float dt) {   // :960-967
    LAUNCH(h, KC_POSITION_CHANGE, k_position_change_b, dim3(list_grids(h).active), dim3(BRICK_THREADS), h->bg, LIST(h, active), (const int8_t*)h->marker,
           (const float*)h->pressure[1], dt, h->vel[0], h->vel[1], h->vel[2]);
    return stage_extrapolate(h);
}
// `step_done`: also publish the number of the step this launch completes (run-ahead throttle of blub_fluid_step)
// Inside blub_fluid_step the kernel also marks the FLUID bricks for the NEXT step's first list build (brick_fluid is all zero here: the
// list build after advection consumed and cleared it), one launch less per step; every entry point that changes particles drops the marks.
static int stage_correct(blub_fluid* h, bool step_done = false) {   // :969-973
    const bool mark = step_done && h->num_ghost == 0;
    if (h->num_particles) {
#define BLUB_CORRECT(F) LAUNCH(h, KC_CORRECT, k_correct<F>, dim3(particle_blocks(h->num_particles)), dim3(256), h->g, h->num_particles, h->pos, h->marker, h->vel[0], h->vel[1], h->vel[2], \
               step_done ? (volatile uint32_t*)h->steps_done_dev : (volatile uint32_t*)nullptr, h->steps_enqueued + 1u, mark ? h->brick_fluid : (uint8_t*)nullptr, h->bg.nbx, h->bg.nby, \
               (const uint32_t*)h->n_dev, N_OWN)
        if (h->filter_mode == BLUB_FILTER_WEIGHTED) BLUB_CORRECT(1); else if (h->filter_mode == BLUB_FILTER_WEIGHTED8) BLUB_CORRECT(2); else BLUB_CORRECT(0);
#undef BLUB_CORRECT
        h->bricks_premarked = mark;
    }
    else if (step_done)
        hipLaunchKernelGGL(k_step_done, dim3(1), dim3(1), 0, h->stream, (volatile uint32_t*)h->steps_done_dev, h->steps_enqueued + 1u);
    return BLUB_OK;
}

// `standalone`: called through blub_fluid_run_stage (test hook) -- the brick lists are then derived from the marker
// volume with every brick active, i.e. the stage has the reference's dense semantics on whatever state was written.
static int run_stage(blub_fluid* h, int stage, float dt, bool standalone) {
    int rc;
    h->cur_stage = BLUB_STAGE_COUNT;
    if (standalone && ((rc = restore_canonical_order(h)) != BLUB_OK || (rc = drop_brick_marks(h)) != BLUB_OK)) return rc;
    if (standalone && stage != BLUB_STAGE_TRANSFER && stage != BLUB_STAGE_BINNING)
        if ((rc = build_lists_from_marker(h)) != BLUB_OK) return rc;
    h->cur_stage = stage;
    h->standalone_stage = standalone;
    if (h->divergence_deferred && stage != BLUB_STAGE_SOLVE_VELOCITY && (rc = stage_divergence(h)) != BLUB_OK) return rc;     // (not reached by blub_fluid_step's order)
    switch (stage) {
    case BLUB_STAGE_TRANSFER: return stage_transfer(h, dt);
    case BLUB_STAGE_DIVERGENCE: return stage_divergence(h, !standalone || h->fuse_divergence == 2);
    case BLUB_STAGE_SOLVE_VELOCITY: return stage_solve(h, 0, dt, standalone);
    case BLUB_STAGE_BINNING: return stage_binning(h);
    case BLUB_STAGE_PROJECT: return stage_project(h);
    case BLUB_STAGE_ADVECT: return stage_advect(h, dt);
    case BLUB_STAGE_DENSITY_GATHER: return stage_density_gather(h, dt);
    case BLUB_STAGE_SOLVE_DENSITY: return stage_solve(h, 1, dt, standalone);
    case BLUB_STAGE_POSITION_CHANGE: return stage_position_change(h, dt);
    case BLUB_STAGE_CORRECT: return stage_correct(h, !standalone);
    }
    return set_error(BLUB_ERR_INVALID_ARGUMENT, "unknown stage");
}

static int check_launch(blub_fluid* h) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char b[256]; snprintf(b, sizeof b, "kernel launch failed: %s", hipGetErrorString(e)); return set_error(BLUB_ERR_DEVICE, b); }
    (void)h;
    return BLUB_OK;
}

static void destroy(blub_fluid* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    F(h->pos); F(h->pos_tmp); for (auto p : h->pvel) F(p); F(h->nodes); F(h->solid_alloc); F(h->scan_totals);
    F(h->gather_sums); F(h->gather_stamp); F(h->phase_stamps[0]); F(h->phase_stamps[1]);
    F(h->pid); F(h->pid_tmp); F(h->resort_counters); F(h->resort_starts); F(h->resort_ranks); F(h->resort_cursor);
    for (auto p : h->vol_owned) F(p);
    for (auto p : h->cgbuf_alloc) F(p);
    F(h->brick_flags); F(h->brick_block_counts); F(h->brick_block_ready); F(h->brick_fluid); F(h->brick_active); F(h->brick_touched); F(h->list_fluid); F(h->list_active); F(h->list_reset); F(h->counts);
    if (h->counts_host) (void)hipHostFree(h->counts_host);
    if (h->steps_done_host) (void)hipHostFree((void*)h->steps_done_host);
    F(h->scalar_log[0]); F(h->scalar_log[1]);
    F(h->tail_sync[0]); F(h->tail_sync[1]); F(h->part4); F(h->pcg1_scalars[0]); F(h->pcg1_scalars[1]); F(h->mesh_positions); F(h->mesh_indices);
    F(h->part_sas); F(h->part_sigma[0]); F(h->part_sigma[1]); F(h->part_max); F(h->tile_flags); F(h->ctrl[0]); F(h->ctrl[1]);
    for (int w = 0; w < 2; ++w) if (h->stats_host[w]) (void)hipHostFree(h->stats_host[w]);
    for (auto& p : h->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : h->prof_pool) (void)hipEventDestroy(e);
    if (h->prof_origin) (void)hipEventDestroy(h->prof_origin);
    if (h->slab) (void)hipFree(h->slab);
    if (h->stream && h->owns_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// Tile geometry of the dense 2.5-D PCG kernels (blub_pcg_dense.hip.h).  0 = the measured default for this grid
// (profiles/r03_dense_sweep_*.txt, all under the deterministic volume placement of vol_alloc): tiles of 512 quads from 256^2-cell planes on
// (8 rows at nx = 256, 4 at nx = 512: y-halo factor 1.25 / 1.5), 256 quads below; as many planes per tile (<= 32: the z-halo factor 1 + 2 / zc)
// as leave >= 4096 waves of tiles for the 256 CUs -- 16 at 256^3 (KD 42.0 / KU 59.2 us against 43.9 / 59.6 with 256-quad tiles), 32 at 512^3
// (KD 328 / KU 545 us against 379 / 539 with 256 x 16).
static void set_dense_geometry(blub_fluid* h, int T, int zc, int grid) {
    PcgGeomZ& gz = h->gz;
    const int qpr = h->g.nx / 4, qpp = qpr * h->g.ny;
    if (T != 256 && T != 512 && T != 1024) T = qpp >= 16384 ? 512 : 256;
    while (T < 2 * qpr && T < 1024) T *= 2;      // a tile holds at least two rows (its halo rows are filled by its first 2 qpr threads)
    gz.g = h->g; gz.qpr = qpr; gz.qpp = qpp; gz.T = T; gz.plane_tiles = (qpp + T - 1) / T;
    if (zc <= 0) { zc = 32; while (zc > 2 && (size_t)gz.plane_tiles * (size_t)((h->g.nz + zc - 1) / zc) * (size_t)(T / 64) < 4096) zc >>= 1; }
    gz.zc = std::max(2, zc);
    gz.z_chunks = (h->g.nz + gz.zc - 1) / gz.zc; gz.tiles = gz.plane_tiles * gz.z_chunks;
    // measured (profiles/r03_dense_march_direction.txt): beyond the Infinity Cache (512^3) the direction kernel gains 0.6 % from alternating and the update
    // kernel loses 4.7 %; at 256^3 alternating the update kernel is the best of the four combinations by ~1 %
    gz.alternate_march = h->dense_alternate_march >= 0 ? h->dense_alternate_march : (h->N >= ((size_t)1 << 26) ? 1 : 2);
    if (grid <= 0) grid = 2048;
    h->pcg_grid_z = std::min(std::min(grid, PCG_GRID_MAX), ((gz.tiles + 7) / 8) * 8);
    gz.flag_factor = 1; gz.flag_chunks = gz.z_chunks;
    // The direction kernel reads r and s with two z-halo planes per tile (1.10x its algorithmic bytes at 32 planes); the update kernel has no halo on
    // p and r and loses from deeper tiles (fewer workgroups).  Beyond the Infinity Cache the direction kernel therefore marches several of the update
    // kernel's chunks per tile, as many as leave 4096 waves of tiles (profiles/r04_dense_sweep.txt, 512^3: KD 310.7 us with 32 planes, 317.7 with 64,
    // 302.4 with 128; KU 502 / 531 / 527 us).
    int f = h->dense_kd_chunk_factor;
    if (f <= 0) { f = 1; if (h->N >= ((size_t)1 << 26)) while (f < 4 && (size_t)gz.plane_tiles * (size_t)((gz.z_chunks + 2 * f - 1) / (2 * f)) * (size_t)(T / 64) >= 4096) f *= 2; }
    PcgGeomZ& gd = h->gzd;
    gd = gz;
    gd.zc = gz.zc * f; gd.z_chunks = (h->g.nz + gd.zc - 1) / gd.zc; gd.tiles = gd.plane_tiles * gd.z_chunks;
    gd.flag_factor = f; gd.flag_chunks = gz.z_chunks;
}

static int create(const blub_fluid_desc* d, blub_fluid** out, hipStream_t shared_stream = nullptr, int vol_z0 = 0, int vol_planes = 0, int mem_mode = 0) {
    if (!d || !out) return set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (d->nx < 4 || d->ny < 3 || d->nz < 3) return set_error(BLUB_ERR_INVALID_ARGUMENT, "grid too small");
    if (d->nx % 4 != 0) return set_error(BLUB_ERR_UNSUPPORTED, "grid_dimension.x must be a multiple of 4 (float4 rows)");
    const uint64_t N64 = (uint64_t)d->nx * d->ny * d->nz;
    if (N64 <= 16384) return set_error(BLUB_ERR_UNSUPPORTED, "grid must have more than 16384 cells (pressure_solver.rs:551)");
    if (N64 >= (1ull << 31)) return set_error(BLUB_ERR_UNSUPPORTED, "grid must have fewer than 2^31 cells");
    if (d->precond_mode > BLUB_PRECOND_LOD0 || d->binning_mode > BLUB_BINNING_OFF)
        return set_error(BLUB_ERR_INVALID_ARGUMENT, "bad quirk mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_error(BLUB_ERR_NO_DEVICE, "no HIP device (libblubhip has no CPU fallback)");
    int dev = d->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return set_error(BLUB_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(dev));
    blub_fluid* h = new (std::nothrow) blub_fluid();
    if (!h) return set_error(BLUB_ERR_OUT_OF_MEMORY, "host allocation failed");
    h->device = dev;
    h->g = Grid{(int)d->nx, (int)d->ny, (int)d->nz};
    h->slab_z0 = 0; h->slab_z1 = (int)d->nz + BZ;   // everything is "own" unless a slab group narrows it
    h->N = (size_t)N64;
    h->vol_z0 = vol_planes > 0 ? vol_z0 : 0; h->vol_planes = vol_planes > 0 ? vol_planes : (int)d->nz;
    h->vol_cells = (size_t)d->nx * d->ny * (size_t)h->vol_planes; h->vol_first = (size_t)d->nx * d->ny * (size_t)h->vol_z0;
    h->max_particles = d->max_num_particles;
    h->precond_mode = d->precond_mode; h->binning_mode = d->binning_mode;
    h->mem_mode = mem_mode;
    int rc = BLUB_OK;
    auto A = [&](int r) { if (rc == BLUB_OK) rc = r; };
    const size_t P = std::max<size_t>(h->max_particles, 1);
    if (shared_stream) { h->stream = shared_stream; h->owns_stream = false; }
    else if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { destroy(h); return set_error(BLUB_ERR_DEVICE, "hipStreamCreate failed"); }
    A(dev_alloc_zero(h->stream, &h->pos, P)); A(dev_alloc_zero(h->stream, &h->pos_tmp, P));
    for (int c = 0; c < 3; ++c) A(dev_alloc_zero(h->stream, &h->pvel[c], P));
    A(dev_alloc_zero(h->stream, &h->nodes, 3 * P)); h->node_stride = (uint32_t)P;
    A(dev_alloc_zero(h->stream, &h->resort_cursor, 4));
    if (d->volume_shift_kib != 0xFFFFFFFFu) {      // (0xFFFFFFFF: one allocation per volume)
        h->slab_shift = (size_t)(d->volume_shift_kib ? d->volume_shift_kib : 64u) * 1024u;
        const size_t per = ((h->vol_cells * 4 + 0x1FFFFFull) & ~0x1FFFFFull) + 0x400000ull;      // 2 MiB alignment + up to 2 MiB of shift
        h->slab_bytes = 18 * per;                                                          // 16 volumes (two of them bytes) + head room
        if (shared_malloc((void**)&h->slab, h->slab_bytes, mem_mode) != hipSuccess) { h->slab = nullptr; h->slab_bytes = 0; (void)hipGetLastError(); }   // (fall back to separate allocations)
    }
    A(vol_alloc(h, &h->residual)); A(vol_alloc(h, &h->search)); A(vol_alloc(h, &h->aux)); A(vol_alloc(h, &h->aux_temp));
    for (int w = 0; w < 2; ++w) A(vol_alloc(h, &h->pressure[w]));
    A(vol_alloc(h, &h->dvol));
    A(vol_alloc(h, &h->marker));
    for (int c = 0; c < 3; ++c) { A(vol_alloc(h, &h->ll[c])); A(vol_alloc(h, &h->vel[c])); }
    A(dev_alloc_zero(h->stream, &h->scan_totals, (h->vol_cells + SCAN_BLOCK - 1) / SCAN_BLOCK + 1));
    PcgGeom& gm = h->geom;
    gm.g = h->g; gm.qpr = h->g.nx / 4; gm.qpp = gm.qpr * h->g.ny; gm.plane_blocks = (gm.qpp + 255) / 256;
    gm.zc = 8;   // LOD0 reading (literal kernel sequence, dense rows): 8 planes per tile, 8 blocks of 256 threads per CU
    gm.z_chunks = (h->g.nz + gm.zc - 1) / gm.zc; gm.tiles = gm.plane_blocks * gm.z_chunks;
    h->pcg_grid = std::min(PCG_GRID_DENSE, gm.tiles);
    set_dense_geometry(h, 0, 0, 0);
    A(dev_alloc_zero(h->stream, &h->part_sas, PCG_GRID_MAX)); A(dev_alloc_zero(h->stream, &h->part_sigma[0], 2 * PCG_GRID_MAX)); A(dev_alloc_zero(h->stream, &h->part_sigma[1], PCG_GRID_MAX));
    {   // tile flags: one per dense tile of ANY tile geometry blub_fluid_set_tuning can select (the smallest tile: 256 quads x 2 planes)
        const size_t max_tiles = (size_t)((gm.qpp + 255) / 256) * (size_t)((h->g.nz + 1) / 2);
        A(dev_alloc_zero(h->stream, &h->part_max, PCG_GRID_MAX)); A(dev_alloc_zero(h->stream, &h->tile_flags, std::max<size_t>(std::max<size_t>(gm.tiles, max_tiles), (size_t)h->gz.tiles) + 8));
    }
    A(dev_alloc_zero(h->stream, &h->ctrl[0], 1)); A(dev_alloc_zero(h->stream, &h->ctrl[1], 1));
    A(dev_alloc_zero(h->stream, &h->tail_sync[0], 1)); A(dev_alloc_zero(h->stream, &h->tail_sync[1], 1));
    {   // the tail kernel's grid barrier needs every block resident at once: bound its grid by what the device holds of THAT kernel
        // (round-2 ADVICE: the bound used to come from another kernel with a smaller footprint).  tail_grid_max = every co-resident block, but
        // the default stays at one block per CU: with ~1000 blocks on the one barrier counter an iteration inside the tail costs 3x as much
        // (measured, round 3: 483-510 steps/s against 929-953 with the tail forced in after 6 iterations)
        int per_cu = 0, per_cu0 = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg1_tail_s<true>, PCG_B_THREADS, 0) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu0, k_pcg_tail_s, PCG_B_THREADS, 0) == hipSuccess && per_cu0 > 0 &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus >= 8) {
            per_cu = std::min(per_cu, per_cu0);
            h->tail_grid_max = ((per_cu * cus) / 8) * 8;
            h->tail_grid = std::min(h->tail_grid_max, (std::min(256, cus) / 8) * 8);
        } else h->use_tail = false;
    }
    BrickGeom& bg = h->bg;
    bg.g = h->g; bg.nbx = (h->g.nx + BX - 1) / BX; bg.nby = (h->g.ny + BY - 1) / BY; bg.nbz = (h->g.nz + BZ - 1) / BZ; bg.nb = bg.nbx * bg.nby * bg.nbz;
    brick_geom_set_magic(bg);
    if ((unsigned long long)bg.nb * (unsigned long long)std::max(bg.nbx, bg.nby) >= 0x100000000ull) A(set_error(BLUB_ERR_UNSUPPORTED, "grid too large for the brick index arithmetic (brick_coords)"));
    h->brick_grid = std::min(bg.nb, BRICK_GRID_MAX);
    A(dev_alloc_zero(h->stream, &h->brick_fluid, (size_t)bg.nb)); A(dev_alloc_zero(h->stream, &h->brick_active, (size_t)bg.nb)); A(dev_alloc_zero(h->stream, &h->brick_touched, (size_t)bg.nb));
    A(dev_alloc_zero(h->stream, &h->list_fluid, (size_t)bg.nb)); A(dev_alloc_zero(h->stream, &h->list_active, (size_t)bg.nb)); A(dev_alloc_zero(h->stream, &h->list_reset, (size_t)bg.nb));
    A(dev_alloc_zero(h->stream, &h->counts, 1));
    A(dev_alloc_zero(h->stream, &h->brick_flags, (size_t)bg.nb)); A(dev_alloc_zero(h->stream, &h->brick_block_counts, (size_t)(bg.nb + 1023) / 1024));
    A(dev_alloc_zero(h->stream, &h->brick_block_ready, (size_t)(bg.nb + 1023) / 1024));
    if (hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) h->num_cus = 0;
    if (rc == BLUB_OK) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&h->steps_done_dev, hp, 0) != hipSuccess) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
        else { memset(hp, 0, 64); h->steps_done_host = (volatile uint32_t*)hp; }
    }
    if (rc == BLUB_OK) {
        // (one entry beyond the ring: its pad0 is the STICKY time-out mark of k_bricks_build -- a ring slot is overwritten 32 builds later,
        //  possibly before the host has looked at it: round-3 ADVICE)
        if (hipHostMalloc((void**)&h->counts_host, (COUNTS_RING + 1) * sizeof(BrickCounts), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&h->counts_host_dev, h->counts_host, 0) != hipSuccess) rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed");
        else memset(h->counts_host, 0, (COUNTS_RING + 1) * sizeof(BrickCounts));
    }
    for (int w = 0; w < 2 && rc == BLUB_OK; ++w) {
        if (hipHostMalloc((void**)&h->stats_host[w], STATS_RING * sizeof(PcgCtrl), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&h->stats_host_dev[w], h->stats_host[w], 0) != hipSuccess) { rc = set_error(BLUB_ERR_OUT_OF_MEMORY, "hipHostMalloc failed"); break; }
        memset(h->stats_host[w], 0, STATS_RING * sizeof(PcgCtrl));
    }
    if (rc != BLUB_OK) { std::string keep = g_last_error; destroy(h); g_last_error = keep; return rc; }
    // the marker volume starts in its static pattern (AIR + SOLID shell): the brick kernels only maintain it locally
    hipLaunchKernelGGL(k_static_marker_dense, dim3(stream_blocks(h->vol_cells / 4)), dim3(256), 0, h->stream, h->g, (const float4*)nullptr, h->marker, h->vol_z0, std::min(h->vol_z0 + h->vol_planes, h->g.nz));
    if (hipStreamSynchronize(h->stream) != hipSuccess) { destroy(h); return set_error(BLUB_ERR_DEVICE, "marker initialisation failed"); }
    *out = h;
    return BLUB_OK;
}

static int poll_stats(blub_fluid* h, bool wait, bool report = true) {   // retrieve_new_error_samples, pressure_solver.rs:148-174
    if (wait) HIP_TRY(hipStreamSynchronize(h->stream));
    for (int w = 0; w < 2; ++w) {
        while (!h->stats_pending[w].empty()) {
            const PendingStat ps = h->stats_pending[w].front();
            const volatile PcgCtrl* c = &h->stats_host[w][ps.slot];
            const uint32_t landed = c->seq;
            if (landed != ps.seq) {
                if ((int32_t)(landed - ps.seq) > 0) { h->stats_pending[w].pop_front(); h->stats_dt[w].pop_front(); continue; }   // slot reused by a newer solve: sample lost
                break;   // this snapshot has not landed yet (samples arrive in order)
            }
            const float max_err = c->max_err, iters = c->num_iter;
            if (c->seq != ps.seq) continue;
            blub_solver_stats s; s.error = max_err * h->stats_dt[w].front(); s.iteration_count = (int32_t)iters;   // :162-163
            h->stats_pending[w].pop_front(); h->stats_dt[w].pop_front();
            if (s.iteration_count < 0) {
                // the persistent tail kernel gave up on a grid barrier (its 256 blocks were not co-resident: shared or partitioned
                // device): the pressure field of that solve is unfinished.  Not a statistics sample; reported as a device error by
                // the next blub_fluid_synchronize / update_statistics, and the tail is not used again on this handle.
                h->failed_solves += 1;
                h->use_tail = false;
                continue;
            }
            h->stats_history[w].push_back(s);
            while (h->stats_history[w].size() > STATS_HISTORY) h->stats_history[w].pop_front();
            h->total_iterations += (uint64_t)s.iteration_count;
        }
    }
    if (report && h->failed_solves != h->failed_solves_reported) {      // (report = false: a caller that drops the status must not consume the error)
        h->failed_solves_reported = h->failed_solves;
        return set_error(BLUB_ERR_DEVICE, "a pressure solve did not finish: the persistent tail kernel timed out on a grid barrier (device shared or partitioned?); the tail is now disabled for this handle");
    }
    return BLUB_OK;
}

}  // namespace blub

#include "blub_slab.inc.hip"

#define REQUIRE_HANDLE(h) do { if (!(h)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle"); if (hipSetDevice((h)->device) != hipSuccess) return blub::set_error(BLUB_ERR_DEVICE, "hipSetDevice failed"); } while (0)

extern "C" {

const char* blub_last_error_string(void) { return blub::g_last_error.c_str(); }
const char* blub_version_string(void) { return "blubhip 0.1 (gfx950)"; }
int blub_fluid_create(const blub_fluid_desc* desc, blub_fluid** out) { return blub::create(desc, out); }
void blub_fluid_destroy(blub_fluid* h) { blub::destroy(h); }

int blub_fluid_create_from_scene(const blub_scene_config* sc, int32_t device, blub_fluid** out) {   // scene/mod.rs:109-144
    if (!sc || !out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    blub_fluid_desc d{};
    d.nx = sc->grid_dimension[0]; d.ny = sc->grid_dimension[1]; d.nz = sc->grid_dimension[2];
    d.max_num_particles = sc->max_num_particles; d.device = device;
    int rc = blub::create(&d, out);
    if (rc != BLUB_OK) return rc;
    const float scale = sc->grid_to_world_scale;
    for (uint32_t i = 0; i < sc->num_fluid_cubes && i < BLUB_SCENE_MAX_CUBES; ++i) {
        float mn[3], mx[3];
        for (int k = 0; k < 3; ++k) { mn[k] = sc->cube_min[i][k] / scale; mx[k] = sc->cube_max[i][k] / scale; }   // :134-137
        rc = blub_fluid_add_fluid_cube(*out, mn, mx);
        if (rc != BLUB_OK) { blub::destroy(*out); *out = nullptr; return rc; }
    }
    float gg[3]; for (int k = 0; k < 3; ++k) gg[k] = sc->gravity[k] / scale;   // :139
    blub_fluid_set_gravity_grid(*out, gg);
    return blub_fluid_synchronize(*out);   // :142
}

int blub_fluid_add_fluid_cube(blub_fluid* h, const float mn[3], const float mx[3]) {
    REQUIRE_HANDLE(h);
    { int rc0 = blub::restore_canonical_order(h); if (rc0 == BLUB_OK) rc0 = blub::drop_brick_marks(h); if (rc0 != BLUB_OK) return rc0; }
    if (!mn || !mx) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    const uint32_t dim[3] = {(uint32_t)h->g.nx, (uint32_t)h->g.ny, (uint32_t)h->g.nz};
    uint32_t count = 0, demand = 0;
    int rc = blub::seed_fluid_cube(dim, h->max_particles, h->num_particles, mn, mx, nullptr, 0, &count);
    if (rc != BLUB_OK) return rc;
    // what the cube asked for without the capacity limit: the reference logs error! for the difference (hybrid_fluid.rs:627-633)
    if (blub::seed_fluid_cube(dim, 0xFFFFFFFFu, h->num_particles, mn, mx, nullptr, 0, &demand) != BLUB_OK) demand = count;
    h->last_add_dropped = demand > count ? demand - count : 0u;
    if (count == 0) return BLUB_OK;
    std::vector<float> buf((size_t)count * 4);
    rc = blub::seed_fluid_cube(dim, h->max_particles, h->num_particles, mn, mx, buf.data(), count, &count);
    if (rc != BLUB_OK) return rc;
    { int rc2 = blub::copy_sync(h, h->pos + h->num_particles, buf.data(), (size_t)count * 16, hipMemcpyHostToDevice); if (rc2 != BLUB_OK) return rc2; }   // queue.write_buffer :671
    h->num_particles += count;
    return BLUB_OK;
}
int blub_fluid_set_gravity_grid(blub_fluid* h, const float g[3]) {
    if (!h || !g) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    for (int k = 0; k < 3; ++k) h->gravity[k] = g[k];
    return BLUB_OK;
}
int blub_fluid_run_stage(blub_fluid* h, int stage, float dt) {
    REQUIRE_HANDLE(h);
    if (!(dt > 0.0f)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "simulation delta must be > 0");
    int rc = blub::run_stage(h, stage, dt, true);
    return rc != BLUB_OK ? rc : blub::check_launch(h);
}
int blub_fluid_step(blub_fluid* h, float dt) {   // hybrid_fluid.rs:770-977
    REQUIRE_HANDLE(h);
    if (!(dt > 0.0f)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "simulation delta must be > 0");
    { int rc0 = blub::bricks_timeout_check(h); if (rc0 != BLUB_OK) return rc0; }      // (sticky mark of an earlier step's list build)
    // Bounded run-ahead ("steps in flight"): on ROCm 7.2 / gfx950 a stream with more than ~1000 queued kernel launches
    // executes 3x slower (measured: 1.38 ms/step with <= 4 steps = 692 launches queued, 4.35 ms/step with 6), so the host waits here
    // until step n - max_steps_in_flight has finished.  The wait polls a counter the device writes into pinned host
    // memory: no HIP call, hence no marker packets in the stream.
    if (h->max_steps_in_flight > 0) {
        // keep the launches in flight below ~700 (the cliff sits at ~1024): a step is ~40 launches + 2 per PCG iteration
        const uint32_t est_launches = 40u + 2u * (uint32_t)(h->cfg[0].max_num_iterations + h->cfg[1].max_num_iterations + 2) * (h->precond_mode == BLUB_PRECOND_ZERO ? 1u : 3u);
        const uint32_t allowed = std::max(1u, std::min(h->max_steps_in_flight, 700u / est_launches));
        unsigned spins = 0;
        while ((int32_t)(h->steps_enqueued - *h->steps_done_host) >= (int32_t)allowed) {
            if (++spins > 2000) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
            if (spins > 50000000u) return blub::set_error(BLUB_ERR_DEVICE, "timed out waiting for an earlier step");
        }
    }
    static const int before_binning[] = {BLUB_STAGE_TRANSFER, BLUB_STAGE_DIVERGENCE, BLUB_STAGE_SOLVE_VELOCITY};
    static const int after_binning[] = {BLUB_STAGE_PROJECT, BLUB_STAGE_ADVECT, BLUB_STAGE_DENSITY_GATHER, BLUB_STAGE_SOLVE_DENSITY, BLUB_STAGE_POSITION_CHANGE, BLUB_STAGE_CORRECT};
    int rc;
    for (int s : before_binning) if ((rc = blub::run_stage(h, s, dt, false)) != BLUB_OK) return rc;
    if (h->rebin_freq != 0 && h->step_counter % h->rebin_freq == 0) {   // :854-856 (Q13)
        if ((rc = blub::run_stage(h, BLUB_STAGE_BINNING, dt, false)) != BLUB_OK) return rc;
        if (h->order_internal && h->binning_mode != BLUB_BINNING_OFF && h->num_particles) {      // the reference's rebinning DEFINES the caller's order anew: what the arrays hold now is it
            hipLaunchKernelGGL(blubk::k_iota, dim3(blub::particle_blocks(h->max_particles)), dim3(256), 0, h->stream, 0u, h->max_particles, h->pid);
            h->order_internal = false;
        }
    } else if (blub::resort_allowed(h) && h->step_counter % (uint32_t)h->resort_every == 0) {      // the engine's own re-sort: same point of the step, positions only
        h->cur_stage = BLUB_STAGE_BINNING;
        if ((rc = blub::stage_resort(h)) != BLUB_OK) return rc;
    }
    for (int s : after_binning) if ((rc = blub::run_stage(h, s, dt, false)) != BLUB_OK) return rc;
    h->step_counter += 1;   // :976
    h->steps_enqueued += 1;   // (k_correct, the last kernel of the step, publishes this number: stage_correct)
    (void)blub::poll_stats(h, false, false);   // the reference polls old read-backs inside solve (:612); an unfinished solve stays pending for synchronize / update_statistics
    return blub::check_launch(h);
}
int blub_fluid_update_statistics(blub_fluid* h) { REQUIRE_HANDLE(h); return blub::poll_stats(h, false); }
int blub_fluid_synchronize(blub_fluid* h) {
    REQUIRE_HANDLE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    { int rc = blub::bricks_timeout_check(h); if (rc != BLUB_OK) return rc; }
    return blub::poll_stats(h, true);
}
int blub_fluid_set_solver_config(blub_fluid* h, int which, const blub_solver_config* cfg) {
    if (!h || !cfg || which < 0 || which > 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    if (cfg->max_num_iterations < 0 || !(cfg->error_tolerance >= 0.0f)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad solver config");
    h->cfg[which] = *cfg;
    return BLUB_OK;
}
int blub_fluid_get_solver_config(const blub_fluid* h, int which, blub_solver_config* cfg) {
    if (!h || !cfg || which < 0 || which > 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    *cfg = h->cfg[which];
    return BLUB_OK;
}
int blub_fluid_solver_stats_count(const blub_fluid* h, int which) { return (!h || which < 0 || which > 1) ? BLUB_ERR_INVALID_ARGUMENT : (int)h->stats_history[which].size(); }
int blub_fluid_solver_stats_get(const blub_fluid* h, int which, int index, blub_solver_stats* out) {
    if (!h || !out || which < 0 || which > 1 || index < 0 || index >= (int)h->stats_history[which].size()) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    *out = h->stats_history[which][index];
    return BLUB_OK;
}
int blub_fluid_solver_stats_latest(const blub_fluid* h, int which, blub_solver_stats* out) {
    if (!h || !out || which < 0 || which > 1 || h->stats_history[which].empty()) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "no statistics sample available");
    *out = h->stats_history[which].back();
    return BLUB_OK;
}
int blub_fluid_set_rebinning_frequency(blub_fluid* h, uint32_t f) { if (!h) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle"); h->rebin_freq = f; return BLUB_OK; }
uint32_t blub_fluid_get_rebinning_frequency(const blub_fluid* h) { return h ? h->rebin_freq : 0; }
uint32_t blub_fluid_num_particles(const blub_fluid* h) { return h ? h->num_particles : 0; }
uint32_t blub_fluid_last_add_dropped(const blub_fluid* h) { return h ? h->last_add_dropped : 0; }
uint32_t blub_fluid_max_num_particles(const blub_fluid* h) { return h ? h->max_particles : 0; }
int blub_fluid_grid_dimension(const blub_fluid* h, uint32_t d[3]) {
    if (!h || !d) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    d[0] = h->g.nx; d[1] = h->g.ny; d[2] = h->g.nz;
    return BLUB_OK;
}
uint32_t blub_fluid_step_counter(const blub_fluid* h) { return h ? h->step_counter : 0; }
int blub_fluid_set_step_counter(blub_fluid* h, uint32_t c) { if (!h) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle"); h->step_counter = c; return BLUB_OK; }
uint64_t blub_fluid_total_solver_iterations(const blub_fluid* h) { return h ? h->total_iterations : 0; }

int blub_fluid_get_device_views(const blub_fluid* h, blub_device_views* v) {
    if (!h || !v) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    v->particles_position_ll = h->pos; v->particles_velocity_x = h->pvel[0]; v->particles_velocity_y = h->pvel[1]; v->particles_velocity_z = h->pvel[2];
    v->velocity_x = h->vel[0]; v->velocity_y = h->vel[1]; v->velocity_z = h->vel[2]; v->marker = h->marker;
    v->pressure_from_velocity = h->pressure[0]; v->pressure_from_density = h->pressure[1]; v->stream = (void*)h->stream;
    return BLUB_OK;
}
int blub_fluid_set_solid_voxels(blub_fluid* h, const float* vox) {
    REQUIRE_HANDLE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!vox) { if (h->solid) { (void)hipFree(h->solid_alloc); h->solid_alloc = nullptr; h->solid = nullptr; } }
    else {
        { int rc2 = blub::solid_ensure(h); if (rc2 != BLUB_OK) return rc2; }
        const size_t held = (size_t)(std::min(h->vol_z0 + h->vol_planes, h->g.nz) - h->vol_z0) * (h->N / (size_t)h->g.nz);
        { int rc2 = blub::copy_sync(h, h->solid_alloc, vox + 4 * h->vol_first, held * sizeof(float4), hipMemcpyHostToDevice); if (rc2 != BLUB_OK) return rc2; }
    }
    // the static marker pattern changed everywhere; every brick may now differ from it
    hipLaunchKernelGGL(blubk::k_static_marker_dense, dim3(blub::stream_blocks(h->vol_cells / 4)), dim3(256), 0, h->stream, h->g, (const float4*)h->solid, h->marker, h->vol_z0, std::min(h->vol_z0 + h->vol_planes, h->g.nz));
    h->all_touched = true;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return BLUB_OK;
}
int blub_fluid_set_meshes(blub_fluid* h, uint32_t nv, const float* positions, uint32_t ni, const uint32_t* indices) {
    REQUIRE_HANDLE(h);
    if ((nv && !positions) || (ni && !indices) || ni % 3 != 0) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad mesh arrays");
    for (uint32_t k = 0; k < ni; ++k) if (indices[k] >= nv) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "mesh index out of range");
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->mesh_positions) { (void)hipFree(h->mesh_positions); h->mesh_positions = nullptr; }
    if (h->mesh_indices) { (void)hipFree(h->mesh_indices); h->mesh_indices = nullptr; }
    h->mesh_num_vertices = nv; h->mesh_num_indices = ni;
    if (nv) { HIP_TRY(hipMalloc((void**)&h->mesh_positions, (size_t)nv * 3 * sizeof(float))); int rc2 = blub::copy_sync(h, h->mesh_positions, positions, (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice); if (rc2 != BLUB_OK) return rc2; }
    if (ni) { HIP_TRY(hipMalloc((void**)&h->mesh_indices, (size_t)ni * sizeof(uint32_t))); int rc2 = blub::copy_sync(h, h->mesh_indices, indices, (size_t)ni * sizeof(uint32_t), hipMemcpyHostToDevice); if (rc2 != BLUB_OK) return rc2; }
    return BLUB_OK;
}
// SceneVoxelization::update, scene/voxelization.rs:116-157 (asynchronous, like the reference's render pass)
int blub_fluid_voxelize(blub_fluid* h, uint32_t num_meshes, const blub_mesh_desc* meshes) {
    REQUIRE_HANDLE(h);
    if (num_meshes && !meshes) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    for (uint32_t m = 0; m < num_meshes; ++m)
        if (meshes[m].index_begin > meshes[m].index_end || meshes[m].index_end > h->mesh_num_indices) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "mesh index range outside the uploaded index buffer");
    { int rc2 = blub::solid_ensure(h); if (rc2 != BLUB_OK) return rc2; }
    {
        blub::ProfScope ps(h, blub::KC_VOXELIZE);
        HIP_TRY(hipMemsetAsync(h->solid_alloc, 0, h->vol_cells * sizeof(float4), h->stream));   // encoder.clear_texture, :123
        for (uint32_t m = 0; m < num_meshes; ++m) {
            static_assert(sizeof(blubk::MeshDesc) == sizeof(blub_mesh_desc), "MeshDesc mirrors blub_mesh_desc");
            blubk::MeshDesc d; memcpy(&d, &meshes[m], sizeof d);
            const uint32_t ntri = (d.index_end - d.index_begin) / 3;
            if (ntri) hipLaunchKernelGGL(blubk::k_voxelize_mesh, dim3((ntri + 3) / 4, blubk::VOXELIZE_SPLIT), dim3(256), 0, h->stream, h->g, d, (const float*)h->mesh_positions, (const uint32_t*)h->mesh_indices, h->solid, h->vol_z0, std::min(h->vol_z0 + h->vol_planes, h->g.nz));
        }
        // the static marker pattern changed: every brick may now differ from it (same as blub_fluid_set_solid_voxels)
        hipLaunchKernelGGL(blubk::k_static_marker_dense, dim3(blub::stream_blocks(h->vol_cells / 4)), dim3(256), 0, h->stream, h->g, (const float4*)h->solid, h->marker, h->vol_z0, std::min(h->vol_z0 + h->vol_planes, h->g.nz));
    }
    h->all_touched = true;
    return blub::check_launch(h);
}
int blub_fluid_set_particles(blub_fluid* h, uint32_t n, const float* pos_ll, const float* vx, const float* vy, const float* vz) {
    REQUIRE_HANDLE(h);
    if (n > h->max_particles) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "more particles than max_num_particles");
    { int rc0 = blub::restore_canonical_order(h); if (rc0 == BLUB_OK) rc0 = blub::drop_brick_marks(h); if (rc0 != BLUB_OK) return rc0; }      // (arrays the caller leaves alone -- null pointers -- stay its own, in its order)
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->num_particles = n; h->have_last_counts = false;      // (brick counts of the old particle set are no estimate for the new one)
    if (n == 0) return BLUB_OK;
    if (pos_ll) { int rc2 = blub::copy_sync(h, h->pos, pos_ll, (size_t)n * 16, hipMemcpyHostToDevice); if (rc2 != BLUB_OK) return rc2; }
    const float* src[3] = {vx, vy, vz};
    for (int c = 0; c < 3; ++c) {
        if (src[c]) { int rc2 = blub::copy_sync(h, h->pvel[c], src[c], (size_t)n * 16, hipMemcpyHostToDevice); if (rc2 != BLUB_OK) return rc2; }
        else { HIP_TRY(hipMemsetAsync(h->pvel[c], 0, (size_t)n * 16, h->stream)); HIP_TRY(hipStreamSynchronize(h->stream)); }
    }
    return BLUB_OK;
}
int blub_fluid_get_particles(blub_fluid* h, float* pos_ll, float* vx, float* vy, float* vz) {
    REQUIRE_HANDLE(h);
    { int rc0 = blub::restore_canonical_order(h); if (rc0 != BLUB_OK) return rc0; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t b = (size_t)h->num_particles * 16;
    if (b == 0) return BLUB_OK;
    if (pos_ll) { int rc2 = blub::copy_sync(h, pos_ll, h->pos, b, hipMemcpyDeviceToHost); if (rc2 != BLUB_OK) return rc2; }
    float* dst[3] = {vx, vy, vz};
    for (int c = 0; c < 3; ++c) if (dst[c]) { int rc2 = blub::copy_sync(h, dst[c], h->pvel[c], b, hipMemcpyDeviceToHost); if (rc2 != BLUB_OK) return rc2; }
    return BLUB_OK;
}
static void* volume_ptr(const blub_fluid* h, int which, size_t* bytes) {
    const size_t N = h->N;
    switch (which) {
    case BLUB_VOLUME_MARKER: *bytes = N; return h->marker;
    case BLUB_VOLUME_LINKED_LIST: *bytes = N * 4; return h->ll[0];
    case BLUB_VOLUME_VELOCITY_X: case BLUB_VOLUME_VELOCITY_Y: case BLUB_VOLUME_VELOCITY_Z: *bytes = N * 4; return h->vel[which - BLUB_VOLUME_VELOCITY_X];
    case BLUB_VOLUME_PRESSURE_VELOCITY: case BLUB_VOLUME_PRESSURE_DENSITY: *bytes = N * 4; return h->pressure[which - BLUB_VOLUME_PRESSURE_VELOCITY];
    case BLUB_VOLUME_RESIDUAL: *bytes = N * 4; return h->residual;
    case BLUB_VOLUME_SEARCH: *bytes = N * 4; return h->search;
    case BLUB_VOLUME_AUX: *bytes = N * 4; return h->aux;
    case BLUB_VOLUME_AUX_TEMP: *bytes = N * 4; return h->aux_temp;
    case BLUB_VOLUME_SOLID: *bytes = N * 16; return h->solid;
    }
    *bytes = 0;
    return nullptr;
}
size_t blub_fluid_volume_bytes(const blub_fluid* h, int which) { size_t b = 0; if (h) (void)volume_ptr(h, which, &b); return b; }
int blub_fluid_read_volume(blub_fluid* h, int which, void* out) {
    REQUIRE_HANDLE(h);
    if (which == BLUB_VOLUME_LINKED_LIST) { int rc0 = blub::restore_canonical_order(h); if (rc0 != BLUB_OK) return rc0; }      // (list heads are particle indices)
    size_t b; void* p = volume_ptr(h, which, &b);
    if (!p || !out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "volume unavailable");
    // (the whole-grid copy only when the volume really holds planes [0, nz): a slab whose plane COUNT equals nz but whose first plane is not 0 -- every rank
    //  of a group allocates the maximum count -- must take the held-plane path: its pointers are allocation - vol_first; round-4 ADVICE)
    if (h->vol_z0 == 0 && h->vol_planes >= h->g.nz) return blub::copy_sync(h, out, p, b, hipMemcpyDeviceToHost);
    // a z-slab holds planes [vol_z0, vol_z0 + vol_planes) only: the rest of the caller's full-grid array reads as zero
    const size_t elem = b / h->N, plane = h->N / (size_t)h->g.nz, z1 = (size_t)std::min(h->vol_z0 + h->vol_planes, h->g.nz);
    memset(out, 0, b);
    return blub::copy_sync(h, (char*)out + h->vol_first * elem, (char*)p + h->vol_first * elem, (z1 - (size_t)h->vol_z0) * plane * elem, hipMemcpyDeviceToHost);
}
int blub_fluid_write_volume(blub_fluid* h, int which, const void* in) {
    REQUIRE_HANDLE(h);
    if (which == BLUB_VOLUME_SOLID) return blub_fluid_set_solid_voxels(h, (const float*)in);
    if (which == BLUB_VOLUME_LINKED_LIST) { int rc0 = blub::restore_canonical_order(h); if (rc0 != BLUB_OK) return rc0; }
    size_t b; void* p = volume_ptr(h, which, &b);
    if (!p || !in) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "volume unavailable");
    if (h->vol_z0 == 0 && h->vol_planes >= h->g.nz) { int rc2 = blub::copy_sync(h, p, in, b, hipMemcpyHostToDevice); if (rc2 != BLUB_OK) return rc2; }
    else {
        const size_t elem = b / h->N, plane = h->N / (size_t)h->g.nz, z1 = (size_t)std::min(h->vol_z0 + h->vol_planes, h->g.nz);
        int rc2 = blub::copy_sync(h, (char*)p + h->vol_first * elem, (const char*)in + h->vol_first * elem, (z1 - (size_t)h->vol_z0) * plane * elem, hipMemcpyHostToDevice);
        if (rc2 != BLUB_OK) return rc2;
    }
    h->all_touched = true;   // arbitrary data may now sit outside the active bricks: the next step re-establishes the invariant
    return BLUB_OK;
}
int blub_fluid_mark_pressure_initialised(blub_fluid* h, int which, int init) {
    if (!h || which < 0 || which > 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    h->pressure_initialised[which] = init != 0;
    return BLUB_OK;
}
int blub_fluid_set_pcg_work_mapping(blub_fluid* h, int mode) {
    if (!h || mode < -1 || mode > 2) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    h->force_pcg_path = mode;
    return BLUB_OK;
}
int blub_fluid_set_pcg_schedule(blub_fluid* h, int mode) {
    if (!h || mode < 0 || mode > 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    h->pcg_schedule = mode;
    return BLUB_OK;
}
int blub_fluid_get_pcg_schedule(const blub_fluid* h) { return h ? h->pcg_schedule : BLUB_ERR_INVALID_ARGUMENT; }
int blub_fluid_set_filter_mode(blub_fluid* h, int mode) {
    if (!h || mode < BLUB_FILTER_SEPARABLE || mode > BLUB_FILTER_WEIGHTED8) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    h->filter_mode = mode;
    return BLUB_OK;
}
int blub_fluid_get_filter_mode(const blub_fluid* h) { return h ? h->filter_mode : BLUB_ERR_INVALID_ARGUMENT; }
int blub_fluid_read_scalar_log(blub_fluid* h, int which, float* out, int capacity, int* count_out) {
    REQUIRE_HANDLE(h);
    if (which < 0 || which > 1 || !count_out || (capacity > 0 && !out)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    if (!h->scalar_log[which]) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "the scalar log is off (blub_fluid_set_tuning \"pcg_scalar_log\" 1)");
    std::vector<float> buf(4 * 1024);
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(buf.data(), h->scalar_log[which], buf.size() * sizeof(float), hipMemcpyDeviceToHost));
    int n = 0;
    while (n < 1024) { uint32_t w[4]; memcpy(w, &buf[4 * (size_t)n], sizeof w); if ((w[0] & w[1] & w[2] & w[3]) == 0xFFFFFFFFu) break; ++n; }
    *count_out = n;
    for (int i = 0; i < n && i < capacity; ++i) memcpy(out + 4 * (size_t)i, &buf[4 * (size_t)i], 4 * sizeof(float));
    return BLUB_OK;
}
int blub_fluid_read_phase_stamps(blub_fluid* h, int which, uint64_t* out, int capacity_iterations) {
    REQUIRE_HANDLE(h);
    if (which < 0 || which > 1 || !out || capacity_iterations <= 0) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    if (!h->phase_stamps[which]) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "the phase stamps are off (blub_fluid_set_tuning \"pcg_phase_stamps\" 1)");
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->phase_stamps[which], (size_t)std::min(capacity_iterations, 64) * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return BLUB_OK;
}
int blub_fluid_last_solve_path(const blub_fluid* h, int which, int* schedule, int* mapping) {
    if (!h || which < 0 || which > 1) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    if (h->last_schedule[which] < 0) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "no solve of this kind has been enqueued yet");
    if (schedule) *schedule = h->last_schedule[which];
    if (mapping) *mapping = h->last_mapping[which];
    return BLUB_OK;
}
// Performance knobs and test hooks by name: none changes results beyond the rounding of a dot-product tree; the library never reads the
// environment.  (Benchmarks and sweeps call this; nothing of the HybridFluid surface does.)
int blub_fluid_set_tuning(blub_fluid* h, const char* name, int value) {
    REQUIRE_HANDLE(h);
    if (!name) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    const std::string k(name);
    if (k == "pcg_tail") h->use_tail = value != 0 && h->tail_grid >= 8;
    else if (k == "pcg_tail_grid") h->tail_grid = std::max(8, (std::min(value, h->tail_grid_max) / 8) * 8);
    else if (k == "pcg_tail_first") h->tail_first_forced = value;
    else if (k == "pcg_tail_inject_timeout") h->tail_inject_timeout = value != 0;
    else if (k == "pcg_tail_margin") h->tail_margin_checks = std::max(0, value);
    else if (k == "pcg_launch_grid") h->pcg_grid_forced = std::max(0, value);
    else if (k == "dense_kd_nt") h->dense_kd_nt = value;
    else if (k == "dense_kd_chunk_factor") { h->dense_kd_chunk_factor = std::max(0, value); set_dense_geometry(h, h->gz.T, h->gz.zc, h->pcg_grid_z); }
    else if (k == "dense_alternate_march") { h->dense_alternate_march = value < 0 ? -1 : (value & 3); set_dense_geometry(h, h->gz.T, h->gz.zc, h->pcg_grid_z); }
    else if (k == "list_launch_grid") h->list_grid_forced = value;
    else if (k == "fuse_divergence") h->fuse_divergence = std::max(0, std::min(2, value));
    else if (k == "p2g_compact") h->p2g_compact = value < 0 ? -1 : (value != 0);
    else if (k == "resort_every") h->resort_every = std::max(0, value);
    else if (k == "p2g_own") h->p2g_own = value != 0;
    else if (k == "bricks_two_kernel_build") h->two_kernel_build = value != 0;
    else if (k == "spin_free") {      // no kernel of a step waits for co-resident workgroups any more: the two-kernel list build, every PCG iteration launched (no persistent tail)
        h->two_kernel_build = value != 0;
        h->use_tail = value == 0 && h->tail_grid >= 8;
    }
    else if (k == "pcg1_max_iterations") h->pcg1_max_iterations = std::max(0, value);
    else if (k == "pcg_scalar_log") {
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int w = 0; w < 2; ++w) {
            if (value && !h->scalar_log[w]) { HIP_TRY(hipMalloc((void**)&h->scalar_log[w], 1024 * sizeof(float4))); HIP_TRY(hipMemset(h->scalar_log[w], 0xFF, 1024 * sizeof(float4))); }
            if (!value && h->scalar_log[w]) { (void)hipFree(h->scalar_log[w]); h->scalar_log[w] = nullptr; }
        }
    }
    else if (k == "pcg_phase_stamps") {
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int w = 0; w < 2; ++w) {
            if (value && !h->phase_stamps[w]) { HIP_TRY(hipMalloc((void**)&h->phase_stamps[w], 64 * 8 * sizeof(unsigned long long))); HIP_TRY(hipMemset(h->phase_stamps[w], 0, 64 * 8 * sizeof(unsigned long long))); }
            if (!value && h->phase_stamps[w]) { (void)hipFree(h->phase_stamps[w]); h->phase_stamps[w] = nullptr; }
        }
    }
    else if (k == "dense_tile_quads" || k == "dense_tile_planes" || k == "dense_grid") {
        HIP_TRY(hipStreamSynchronize(h->stream));
        blub::set_dense_geometry(h, k == "dense_tile_quads" ? value : h->gz.T, k == "dense_tile_planes" ? value : (k == "dense_tile_quads" ? 0 : h->gz.zc), k == "dense_grid" ? value : 0);
    } else return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "unknown tuning knob");
    return BLUB_OK;
}
int blub_fluid_set_max_steps_in_flight(blub_fluid* h, uint32_t m) { if (!h) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle"); h->max_steps_in_flight = m; return BLUB_OK; }
int blub_fluid_get_brick_counts(blub_fluid* h, uint32_t out[6]) {
    REQUIRE_HANDLE(h);
    if (!out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipStreamSynchronize(h->stream));
    blubk::BrickCounts bc{};
    { int rc2 = blub::copy_sync(h, &bc, h->counts, sizeof(bc), hipMemcpyDeviceToHost); if (rc2 != BLUB_OK) return rc2; }
    out[0] = bc.n_fluid; out[1] = bc.n_active; out[2] = bc.n_reset; out[3] = bc.n_stale; out[4] = (uint32_t)h->bg.nb; out[5] = blubk::BX * blubk::BY * blubk::BZ;
    return BLUB_OK;
}
int blub_fluid_profile_enable(blub_fluid* h, int enabled) { REQUIRE_HANDLE(h); int rc = blub::prof_flush(h); h->prof_enabled = enabled != 0; return rc; }
int blub_fluid_profile_reset(blub_fluid* h) {
    REQUIRE_HANDLE(h);
    int rc = blub::prof_flush(h);
    for (int k = 0; k < blub::KC_COUNT; ++k) { h->prof_ms[k] = 0; h->prof_launches[k] = 0; }
    h->prof_trace.clear();
    if (h->prof_origin) { h->prof_pool.push_back(h->prof_origin); h->prof_origin = nullptr; }
    return rc;
}
int blub_fluid_profile_trace(blub_fluid* h, blub_trace_event* events, int capacity, int* count_out) {
    REQUIRE_HANDLE(h);
    if (!count_out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    int rc = blub::prof_flush(h);
    if (rc != BLUB_OK) return rc;
    const int n = (int)h->prof_trace.size();
    *count_out = n;
    if (!events) return BLUB_OK;   // size query
    for (int i = 0; i < n && i < capacity; ++i) {
        const auto& t = h->prof_trace[i];
        memset(&events[i], 0, sizeof(blub_trace_event));
        strncpy(events[i].name, blub::kKernelClassNames[t.kc], sizeof(events[i].name) - 1);
        events[i].stage = (uint32_t)t.stage; events[i].step = t.step; events[i].start_us = t.start_us; events[i].duration_us = t.dur_us;
    }
    return BLUB_OK;
}
int blub_fluid_profile_read(blub_fluid* h, blub_prof_entry* entries, int capacity, int* count_out) {
    REQUIRE_HANDLE(h);
    if (!entries || !count_out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    int rc = blub::prof_flush(h);
    if (rc != BLUB_OK) return rc;
    int n = 0;
    for (int k = 0; k < blub::KC_COUNT && n < capacity; ++k) {
        if (!h->prof_launches[k]) continue;
        memset(&entries[n], 0, sizeof(blub_prof_entry));
        strncpy(entries[n].name, blub::kKernelClassNames[k], sizeof(entries[n].name) - 1);
        entries[n].launches = h->prof_launches[k]; entries[n].total_ms = h->prof_ms[k];
        ++n;
    }
    *count_out = n;
    return BLUB_OK;
}
}  // extern "C"
