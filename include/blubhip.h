/* blubhip.h -- C-ABI of the MI355X-native APIC fluid-step engine (libblubhip.so).
 *
 * This is the drop-in boundary for blub's `HybridFluid` (reference: src/simulation/mod.rs:4-5 re-exports
 * `HybridFluid`, `SolverConfig`, `SolverStatisticSample`).  The reference has no FFI layer -- the Rust struct is the
 * boundary -- so every entry point below names the Rust method it replaces (file:line relative to /root/reference).
 * A Rust `HybridFluid` shim binding these symbols is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns BLUB_OK (0) or a negative blub_status and never
 * throws/aborts; one handle = one HIP stream = one host thread at a time (the reference type is !Send);
 * grid space everywhere (cells), x fastest, index = (z*ny + y)*nx + x; all reals f32.
 * `blub_fluid_step` only ENQUEUES work on the handle's stream (the reference only records into the caller's command
 * encoder, hybrid_fluid.rs:770-977); call blub_fluid_synchronize / blub_fluid_update_statistics afterwards.
 */
#ifndef BLUBHIP_H
#define BLUBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct blub_fluid blub_fluid; /* opaque; owns every particle buffer and grid volume (hybrid_fluid.rs:24-72) */

typedef enum blub_status {
    BLUB_OK = 0,
    BLUB_ERR_INVALID_ARGUMENT = -1,
    BLUB_ERR_UNSUPPORTED = -2,      /* e.g. nx not a multiple of 4, N <= 16384 (pressure_solver.rs:551 asserts the same) */
    BLUB_ERR_OUT_OF_MEMORY = -3,
    BLUB_ERR_DEVICE = -4,           /* a HIP call failed; blub_last_error_string() has the text */
    BLUB_ERR_IO = -5,               /* scene file unreadable */
    BLUB_ERR_PARSE = -6,            /* scene JSON malformed / missing field */
    BLUB_ERR_NO_DEVICE = -7,        /* no HIP device: there is NO CPU fallback in this library */
    BLUB_ERR_COMM = -8              /* RCCL failure */
} blub_status;

/* hybrid_fluid.glsl:20-22 (R8Snorm marker values) */
enum { BLUB_CELL_SOLID = 0, BLUB_CELL_FLUID = 1, BLUB_CELL_AIR = -1 };
/* hybrid_fluid.rs:90 */
enum { BLUB_PARTICLES_PER_GRID_CELL = 8 };

typedef enum blub_solver { BLUB_SOLVER_VELOCITY = 0, BLUB_SOLVER_DENSITY = 1 } blub_solver; /* hybrid_fluid.rs:259-260 */

/* pressure_solver.rs:57-62 `SolverConfig` (defaults hybrid_fluid.rs:253-257: 0.1 / 32 / 4) */
typedef struct blub_solver_config {
    float error_tolerance;
    int32_t max_num_iterations;
    int32_t error_check_frequency;
} blub_solver_config;

/* pressure_solver.rs:64-68 `SolverStatisticSample`; error = max|r| * dt (pressure_solver.rs:162) */
typedef struct blub_solver_stats {
    float error;
    int32_t iteration_count;
} blub_solver_stats;

/* Reference quirks that need an explicit switch. Zero-initialised = parity defaults. */
typedef enum blub_precond_mode {
    BLUB_PRECOND_ZERO = 0, /* Q1: neighbour texelFetch at lod 1 returns 0 => z = (r/d)/d   (parity default) */
    BLUB_PRECOND_LOD0 = 1  /* Q1: neighbour fetches clamp to lod 0 (literal two-pass stencil) */
} blub_precond_mode;
typedef enum blub_binning_mode {
    BLUB_BINNING_FIXED = 0, /* Q4: guarded, 0-based permutation (default) */
    BLUB_BINNING_LITERAL = 1, /* Q4 as the reference's shaders run it: no i < NumParticles guard (threads up to ceil(P/64)*64 bin stale / zero
                             * records), 1-based destinations (slot 0 is never written), the whole buffer copied back: every pass replaces
                             * pad + 1 real particles by stale / zero records (particle_binning_count.comp:9-13, _rewrite_particles.comp:8-16, hybrid_fluid.rs:885-891) */
    BLUB_BINNING_OFF = 2    /* never rebin */
} blub_binning_mode;

typedef struct blub_fluid_desc {
    uint32_t nx, ny, nz;           /* grid_dimension  (hybrid_fluid.rs:94) */
    uint32_t max_num_particles;    /* hybrid_fluid.rs:95 */
    int32_t device;                /* HIP device ordinal; -1 = current device */
    uint32_t precond_mode;         /* blub_precond_mode */
    uint32_t binning_mode;         /* blub_binning_mode */
    uint32_t volume_shift_kib;     /* placement of the grid volumes inside the engine's one allocation: volume i is shifted by i x this many KiB against its
                                    * 2 MiB-aligned slot (0 = default 64; 0xFFFFFFFF = one allocation per volume).  A performance knob: see vol_alloc in blub_fluid.hip */
} blub_fluid_desc;

/* ---- scene JSON: src/scene/mod.rs:19-43 (SceneConfig / FluidConfig / Box) -- host only, no device needed -------- */
enum { BLUB_SCENE_MAX_CUBES = 64, BLUB_SCENE_MAX_STATIC_OBJECTS = 16, BLUB_SCENE_MAX_PATH = 256 };
typedef enum blub_animation_curve { BLUB_CURVE_LINEAR = 0, BLUB_CURVE_SMOOTHSTEP = 1 } blub_animation_curve;   /* models.rs:21-25 */
typedef struct blub_static_object {    /* StaticObjectConfig + RigidAnimation, scene/models.rs:11-46 */
    char model[BLUB_SCENE_MAX_PATH];       /* relative to the `models` directory (models.rs:267) */
    float world_position[3];
    float scale;
    float rotation_angles_deg[3];          /* cgmath::Euler<Deg<f32>> */
    uint32_t has_translation;              /* animation.translation (models.rs:27-32) */
    float translation_target[3];
    uint32_t translation_curve;            /* blub_animation_curve */
    float translation_duration;            /* seconds to reach the target */
    uint32_t has_rotation;                 /* animation.rotation (models.rs:34-38) */
    float rotation_axis[3];
    float rotation_deg_per_sec;
} blub_static_object;
typedef struct blub_scene_config {
    float gravity[3];              /* world space */
    float world_position[3];
    float grid_to_world_scale;
    uint32_t grid_dimension[3];
    uint32_t max_num_particles;
    uint32_t num_fluid_cubes;
    float cube_min[BLUB_SCENE_MAX_CUBES][3]; /* world space */
    float cube_max[BLUB_SCENE_MAX_CUBES][3];
    uint32_t num_static_objects;
    blub_static_object static_objects[BLUB_SCENE_MAX_STATIC_OBJECTS];
} blub_scene_config;

int blub_scene_load_json(const char* path, blub_scene_config* out);              /* Scene::new, scene/mod.rs:64-66 */
int blub_scene_parse_json(const char* text, size_t len, blub_scene_config* out);
/* Host-side restatement of HybridFluid::add_fluid_cube's particle generator (hybrid_fluid.rs:609-678): writes
 * `count` ParticlePositionLl records (x,y,z,0xFFFFFFFF) for a cube given in GRID space when the fluid already holds
 * `num_particles_before` particles. *count_out = number generated (truncated to capacity like :627-633). */
int blub_seed_fluid_cube(const uint32_t grid_dim[3], uint32_t max_num_particles, uint32_t num_particles_before,
                         const float min_grid[3], const float max_grid[3], float* pos_ll_out, size_t capacity,
                         uint32_t* count_out);

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
int blub_fluid_create(const blub_fluid_desc* desc, blub_fluid** out);             /* HybridFluid::new, hybrid_fluid.rs:92-607 */
/* Scene::create_fluid_from_config, scene/mod.rs:109-144: new + add_fluid_cube(cube/scale) for every cube +
 * set_gravity_grid(gravity/scale) */
int blub_fluid_create_from_scene(const blub_scene_config* scene, int32_t device, blub_fluid** out);
void blub_fluid_destroy(blub_fluid* h);
const char* blub_last_error_string(void);
const char* blub_version_string(void);

/* ---- HybridFluid surface -------------------------------------------------------------------------------------- */
int blub_fluid_add_fluid_cube(blub_fluid* h, const float min_grid[3], const float max_grid[3]);      /* hybrid_fluid.rs:620 */
int blub_fluid_set_gravity_grid(blub_fluid* h, const float gravity_grid[3]);                         /* :692 */
int blub_fluid_step(blub_fluid* h, float simulation_delta_seconds);                                  /* :770 (dt = Duration::as_secs_f32) */
int blub_fluid_update_statistics(blub_fluid* h);                                                     /* :765; non-blocking poll */
int blub_fluid_synchronize(blub_fluid* h);                                                           /* device.poll(Wait), scene/mod.rs:142 */
int blub_fluid_set_solver_config(blub_fluid* h, int which, const blub_solver_config* cfg);           /* :743-749 */
int blub_fluid_get_solver_config(const blub_fluid* h, int which, blub_solver_config* cfg);
/* pressure_solver_stats_{velocity,density}(): history of <= 100 samples (pressure_solver.rs:101), oldest first */
int blub_fluid_solver_stats_count(const blub_fluid* h, int which);                                   /* :755-761 */
int blub_fluid_solver_stats_get(const blub_fluid* h, int which, int index, blub_solver_stats* out);
int blub_fluid_solver_stats_latest(const blub_fluid* h, int which, blub_solver_stats* out);
int blub_fluid_set_rebinning_frequency(blub_fluid* h, uint32_t every_n_steps);                       /* dynamic_settings(), :751, 19-22 */
uint32_t blub_fluid_get_rebinning_frequency(const blub_fluid* h);
uint32_t blub_fluid_num_particles(const blub_fluid* h);                                              /* :696 */
uint32_t blub_fluid_max_num_particles(const blub_fluid* h);
/* Particles the LAST blub_fluid_add_fluid_cube call could not add because max_num_particles was reached (the call itself returns
 * BLUB_OK and truncates, like the reference, which logs `error!`: hybrid_fluid.rs:627-633). */
uint32_t blub_fluid_last_add_dropped(const blub_fluid* h);
int blub_fluid_grid_dimension(const blub_fluid* h, uint32_t dim_out[3]);                             /* :727 */
uint32_t blub_fluid_step_counter(const blub_fluid* h);
int blub_fluid_set_step_counter(blub_fluid* h, uint32_t c);

/* bind_group_renderer(), hybrid_fluid.rs:700-723 + shader/fluid_render_info.glsl:11-23: device pointers, read-only
 * for the caller.  The volume pointers are valid until destroy.  The four PARTICLE buffers: re-query after a step (the engine's own re-sort writes the positions
 * into a second buffer and swaps: "resort_every"), and between two by-index calls (get / set particles, the linked-list volume, the stage hook) their slots are in
 * the engine's internal order -- a permutation of the caller's, the same for all four buffers, which a point renderer does not see; blub_fluid_get_particles
 * returns the caller's order. */
typedef struct blub_device_views {
    const void* particles_position_ll;  /* float4 {x,y,z, u32 linked_list_next} */
    const void* particles_velocity_x;   /* float4 {C row xyz, v_x}   (particles.glsl:13-15) */
    const void* particles_velocity_y;
    const void* particles_velocity_z;
    const void* velocity_x;             /* f32 volumes, staggered on the positive faces */
    const void* velocity_y;
    const void* velocity_z;
    const void* marker;                 /* int8 volume */
    const void* pressure_from_velocity; /* f32 */
    const void* pressure_from_density;  /* f32 */
    void* stream;                       /* hipStream_t the engine enqueues on */
} blub_device_views;
int blub_fluid_get_device_views(const blub_fluid* h, blub_device_views* out);

/* Stand-in for the borrowed `SceneVoxelization` RGBA16F volume (scene/voxelization.rs:17, hybrid_fluid.rs:266):
 * N float4 {solid velocity xyz, solid flag w} or NULL for the all-zero volume every BASELINE scene has. */
int blub_fluid_set_solid_voxels(blub_fluid* h, const float* voxels_xyzw_or_null);

/* ---- static objects: scene/models.rs + scene/voxelization.rs + shader/voxelize/conservative_hull.{vert,frag} ----
 * The reference rasterises every mesh with the hardware's conservative rasteriser into the RGBA16F volume once per step
 * (scene/mod.rs:192-196).  Here the same per-fragment logic runs as a compute kernel over the pixels whose square
 * overlaps the projected triangle ("overestimate" conservative rasterisation; depth = the triangle's plane at the pixel
 * centre, clamped to the triangle's depth range).  Velocities are rounded to f16 like the RGBA16F store. */
typedef struct blub_mesh_desc {              /* the fields of MeshDataGpu (models.rs:55-70) the voxeliser reads */
    float voxel_transform[3][4];             /* rows of transform_voxel: voxel = (M * vec4(p, 1)).xyz (conservative_hull.vert:17-21) */
    float fluid_space_velocity[3];
    float fluid_space_rotation_axis_scaled[3];
    uint32_t index_begin, index_end;         /* index_buffer_range */
} blub_mesh_desc;
/* StaticMeshData::to_gpu (models.rs:156-228) for object `object_index` of the scene: rigid animation evaluated at
 * `total_simulated_time` (Timer::total_simulated_time, already advanced by the step being taken: timer.rs:124), velocity by
 * the reference's backward difference over `simulation_delta`.  index_begin/index_end are left 0.  Host only. */
int blub_scene_mesh_desc_at_time(const blub_scene_config* scene, uint32_t object_index, uint64_t total_simulated_time_ns,
                                 uint64_t simulation_delta_ns, blub_mesh_desc* out);
/* Stand-in for tobj::load_obj(triangulate) (models.rs:267-276): `v` and `f` records only, polygons fan-triangulated, indices
 * into the position list.  Call with NULL outputs to query the sizes.  Host only. */
int blub_load_obj(const char* path, float* positions_xyz, size_t vertex_capacity, uint32_t* num_vertices, uint32_t* indices,
                  size_t index_capacity, uint32_t* num_indices);
/* SceneModels::from_config's vertex / index buffers (models.rs:354-375); uploaded once. */
int blub_fluid_set_meshes(blub_fluid* h, uint32_t num_vertices, const float* positions_xyz, uint32_t num_indices, const uint32_t* indices);
/* SceneVoxelization::update (voxelization.rs:116-157): clear the solid volume, then one conservative-hull pass per mesh in
 * order (a later mesh overwrites an earlier one).  Enqueues on the handle's stream; the next step sees the new solids. */
int blub_fluid_voxelize(blub_fluid* h, uint32_t num_meshes, const blub_mesh_desc* meshes);

/* ---- state exchange (parity tests, checkpoint/resume) ---------------------------------------------------------- */
int blub_fluid_set_particles(blub_fluid* h, uint32_t n, const float* pos_ll, const float* vx, const float* vy, const float* vz);
int blub_fluid_get_particles(blub_fluid* h, float* pos_ll, float* vx, float* vy, float* vz); /* any may be NULL; blocks */

typedef enum blub_volume {
    BLUB_VOLUME_MARKER = 0,            /* int8   */
    BLUB_VOLUME_LINKED_LIST = 1,       /* uint32 (list heads of component x / density; the binning counters live in AUX_TEMP) */
    BLUB_VOLUME_VELOCITY_X = 2, BLUB_VOLUME_VELOCITY_Y = 3, BLUB_VOLUME_VELOCITY_Z = 4,
    BLUB_VOLUME_PRESSURE_VELOCITY = 5, BLUB_VOLUME_PRESSURE_DENSITY = 6,
    /* RESIDUAL, SEARCH, AUX, AUX_TEMP are solver work volumes: their contents OUTSIDE FLUID cells are undefined (AUX_TEMP doubles as the u32
     * counter / prefix-sum scratch of the binning pass, so non-FLUID cells may hold integer bit patterns, NaNs included; every reader gates on
     * the FLUID bit of the stencil descriptor), and which allocation backs RESIDUAL / SEARCH may change from solve to solve (double buffering
     * by iteration parity) -- re-query blub_fluid_get_device_views / read_volume after a step instead of caching their device pointers. */
    BLUB_VOLUME_RESIDUAL = 7, BLUB_VOLUME_SEARCH = 8, BLUB_VOLUME_AUX = 9, BLUB_VOLUME_AUX_TEMP = 10,
    BLUB_VOLUME_SOLID = 11             /* float4 */
} blub_volume;
size_t blub_fluid_volume_bytes(const blub_fluid* h, int which);
int blub_fluid_read_volume(blub_fluid* h, int which, void* host_out);        /* blocks */
int blub_fluid_write_volume(blub_fluid* h, int which, const void* host_in);  /* blocks */
int blub_fluid_mark_pressure_initialised(blub_fluid* h, int which, int initialised); /* pressure_solver.rs:601-603 */

/* The rows of HybridFluid::step in recorded order; blub_fluid_step == all of them in sequence
 * (+ BINNING every rebinning_frequency steps).  Exposed so that each stage can be compared with the oracle on
 * identical inputs. */
typedef enum blub_stage {
    BLUB_STAGE_TRANSFER = 0,        /* hybrid_fluid.rs:806-833: clear + linked lists + boundary marker + gather x/y/z */
    BLUB_STAGE_DIVERGENCE = 1,      /* :836-840 */
    BLUB_STAGE_SOLVE_VELOCITY = 2,  /* :843-852 */
    BLUB_STAGE_BINNING = 3,         /* :857-893 */
    BLUB_STAGE_PROJECT = 4,         /* :906-914 divergence_remove + extrapolate_velocity */
    BLUB_STAGE_ADVECT = 5,          /* :916-932 clear + advect_particles + boundary marker */
    BLUB_STAGE_DENSITY_GATHER = 6,  /* :933-937 */
    BLUB_STAGE_SOLVE_DENSITY = 7,   /* :940-949 */
    BLUB_STAGE_POSITION_CHANGE = 8, /* :960-967 position_change + extrapolate_velocity */
    BLUB_STAGE_CORRECT = 9,         /* :969-973 */
    BLUB_STAGE_COUNT = 10
} blub_stage;
int blub_fluid_run_stage(blub_fluid* h, int stage, float simulation_delta_seconds);

/* ---- measurement ---------------------------------------------------------------------------------------------- */
/* Per-stage / per-kernel-class device time, measured with hipEvents on the handle's own stream. When profiling is
 * enabled every kernel class launch is bracketed by events (adds host overhead; do not time steps/s with it on). */
enum { BLUB_PROF_MAX_ENTRIES = 48 };
typedef struct blub_prof_entry {
    char name[48];
    uint64_t launches;
    double total_ms;
} blub_prof_entry;
int blub_fluid_profile_enable(blub_fluid* h, int enabled);
int blub_fluid_profile_reset(blub_fluid* h);
int blub_fluid_profile_read(blub_fluid* h, blub_prof_entry* entries, int capacity, int* count_out); /* blocks */
/* Per-launch timeline of the profiled launches since the last reset (the data behind the reference's chrome-trace dump,
 * gui/mod.rs:487-491 / wgpu-profiler): start relative to the first profiled launch, both in microseconds. */
typedef struct blub_trace_event {
    char name[48];        /* kernel class */
    uint32_t stage;       /* blub_stage the launch belongs to (BLUB_STAGE_COUNT = list building / other) */
    uint32_t step;        /* step counter at enqueue time */
    double start_us, duration_us;
} blub_trace_event;
int blub_fluid_profile_trace(blub_fluid* h, blub_trace_event* events, int capacity, int* count_out); /* blocks */
/* Work mapping of the PCG kernels: -1 = automatic (brick lists on grids of up to 1 M cells and while fewer than max(8192, 30 % of the) bricks hold fluid
 * -- max(2048, 20 %) with the reference's two-reduction order --, dense rows otherwise: a measured speed choice, profiles/r03_mapping_crossover.txt),
 * 0 = dense rows, 1 (or 2, its former LDS-staged alias) = brick lists.  A performance knob only: both mappings run the same per-cell arithmetic (the
 * dot-product partial sums are grouped differently, so results agree to rounding, not bitwise).  Within a mapping the grouping depends on the
 * brick lists alone, never on launch grids or host timing: two solves on the same state give bit-identical scalars. */
int blub_fluid_set_pcg_work_mapping(blub_fluid* h, int mode);
/* Schedule of the PCG iteration on the brick mappings: 0 = the reference's (pressure_solver.rs:654-723: two global reductions, here
 * two kernels per iteration), 1 = single-reduction (Chronopoulos-Gear) form of the same recurrence: ONE kernel per iteration, A d
 * carried by a recurrence.  Identical in exact arithmetic, a different rounding in f32 (not bit-comparable); convergence test,
 * check cadence and statistics are the same.  The dense-row mapping always runs schedule 0 (it is byte-, not launch-bound).
 * Default 1 (since round 4; rounds 1-3 shipped 0 and benchmarked 1): the step of the metric's configuration is bound by the number of
 * dependent launches, and the reference's order costs 35 % of the steps/s there.  What the default rests on: fixed-k solves against the
 * oracle AND against the reference's own shaders at the tolerances of schedule 0 (tests/test_gpu_pcg_schedule.py, tests/test_gpu_vs_ref.py),
 * the same loose whole-step bound (0.15 cells), 600-step runs of dam_halfhalf and corner_dams_256 whose iteration-count and reported-error
 * distributions match schedule 0's, and the measured gap between the carried residual and b - A p at the end of those solves
 * (tests/test_gpu_pcg_schedule.py::test_600_steps_..., profiles/r04_schedule_longrun_{dam_halfhalf,corner_dams_256}.json).  Its r and q = A d are carried by recurrences
 * (no residual replacement: the gap stays at rounding level for the <= 64 iterations it is used for); solves configured with more than 64
 * iterations run schedule 0 regardless ("pcg1_max_iterations", blub_fluid_set_tuning), as does the dense-row mapping.
 * blub_fluid_set_pcg_schedule(h, 0) selects the reference's literal order of operations everywhere. */
int blub_fluid_set_pcg_schedule(blub_fluid* h, int mode);
int blub_fluid_get_pcg_schedule(const blub_fluid* h);
/* What the most recently enqueued solve `which` (0 velocity, 1 density) ACTUALLY ran -- the selected schedule is a request: the dense-row mapping, the
 * LOD0 preconditioner reading and solves configured beyond "pcg1_max_iterations" always run the reference's order.  *schedule: 0 = the reference's
 * two-reduction order (pressure_solver.rs:654-723 literally), 1 = single-reduction; *mapping: 0 = dense rows, 1 = brick lists, 2 = the literal kernel
 * sequence of the LOD0 reading.  Either pointer may be NULL.  BLUB_ERR_INVALID_ARGUMENT before the first solve.  (Round-4 ADVICE: a drop-in caller
 * must be able to tell which rounding of the recurrence produced a pressure field.) */
int blub_fluid_last_solve_path(const blub_fluid* h, int which, int* schedule, int* mapping);
/* Arithmetic of the hardware trilinear filter the reference samples with in density_projection_correct_particles.comp:32-40 (R3: the position change) and
 * advect_particles.comp:155-163 (A1: the push out of a solid).  Vulkan leaves it to the implementation; the three evaluations the shim that runs the
 * reference's own shaders offers (oracle/glsl/), each reproduced BIT FOR BIT by the engine in the same mode (tests/test_gpu_vs_ref.py):
 *   BLUB_FILTER_SEPARABLE (default): f32 lerps along x, y, z -- what the oracle evaluates;
 *   BLUB_FILTER_WEIGHTED: the weighted sum of Vulkan 1.2 16.8.3 in f32 (<= 4e-6 cells from SEPARABLE);
 *   BLUB_FILTER_WEIGHTED8: the same with 8-bit filter weights -- what a real sampler does, i.e. the closest to "the reference wgpu path on the host's own
 *     GPU" (<= 3e-3 cells at p99.9 from SEPARABLE in R3).  A quirk switch like the preconditioner reading: it changes results, not speed. */
enum { BLUB_FILTER_SEPARABLE = 0, BLUB_FILTER_WEIGHTED = 1, BLUB_FILTER_WEIGHTED8 = 2 };
int blub_fluid_set_filter_mode(blub_fluid* h, int mode);
int blub_fluid_get_filter_mode(const blub_fluid* h);
/* Diagnostic (blub_fluid_set_tuning "pcg_scalar_log" 1 switches it on): the scalars of the most recent single-reduction solve `which`, 4 floats per
 * iteration i -- {gamma_i = r.u, delta_i = w.u, max|r_i|, alpha_i} exactly as K(i) reduced them from the partial array (pressure_solver.rs:654-723 keeps
 * the same quantities in its 16-float control buffer).  *count_out = iterations that ran (<= 1024); at most `capacity` entries are copied.  Every slab of
 * a z-slab group derives bit-identical scalars, whatever the transport: the direct-transport probe compares the logs between ranks and transports. Blocks. */
int blub_fluid_read_scalar_log(blub_fluid* h, int which, float* out, int capacity, int* count_out);
/* Diagnostic (blub_fluid_set_tuning "pcg_phase_stamps" 1 switches it on): the intra-kernel timeline of the most recent single-reduction solve `which` on
 * the brick mapping.  Workgroup 0 of K(i) leaves eight 64-bit time stamps (s_memrealtime: 100 MHz, 10 ns ticks; 0 = not reached) in entry i < 64:
 * [0] entry, [1] first round trip back (`done`, list length), [2] partials reduced / alpha, beta known, [3] first brick: descriptors and fields arrived,
 * r / u / q / d / p updated, tile written, [4] after the tile barrier, [5] w = A u formed and stored, [6] partial stored (end), [7] unused.
 * `out`: capacity_iterations x 8 values.  Blocks.  (tools/kiter_timeline.py turns it into the table under profiles/.) */
int blub_fluid_read_phase_stamps(blub_fluid* h, int which, uint64_t* out, int capacity_iterations);
/* Performance knobs / test hooks by name (the library never reads the environment).  None changes a result beyond the rounding of a
 * dot-product tree.  "pcg_tail" 0|1: persistent tail kernel of the single-reduction solves; "pcg_tail_first" n: hand over to the tail after
 * exactly n launched iterations (-1: predicted from the last solves); "pcg_tail_margin" n: check intervals launched beyond the prediction;
 * "pcg_tail_inject_timeout" 1 (test hook): the next tail kernel finds its grid barrier timed out -- the solve is reported unfinished
 *   (BLUB_ERR_DEVICE at the next synchronize / update_statistics) and the handle stops using the tail;
 * "pcg1_max_iterations" n (default 64): solves configured with more iterations run schedule 0 even when schedule 1 is selected;
 * "pcg_scalar_log" 0|1: keep the per-iteration scalars of the single-reduction solves for blub_fluid_read_scalar_log (one 16-byte store per launch);
 * "pcg_phase_stamps" 0|1: workgroup 0 of every K(i) leaves time stamps at its phase boundaries for blub_fluid_read_phase_stamps (seven 8-byte stores);
 * "resort_every" n (default 8; 0 = never): the engine re-sorts its particle arrays by (brick, cell) every n-th step, at the point of the step where the
 *   reference rebins (positions only move there).  Every particle kernel is bound by sector requests and follows the order of the particles in memory;
 *   the reference's own cadence (60 steps) is kept for what the CALLER sees -- blub_fluid_get_particles, the linked-list volume and every other
 *   by-index entry point return the caller's order (a particle-id array restores it) -- only the device views' particle buffers are in the engine's
 *   internal order between two such calls (a renderer does not care).  Single domains only; not with BLUB_BINNING_LITERAL;
 * "p2g_own" 0|1 (default 1): well-filled bricks gather list-centrically (every P2G list walked once, brick-boundary faces finished by a second small
 *   kernel: another association of <= 8 partial sums on those faces); 0: always tile-centric;
 * "pcg_launch_grid" n: launch grid of the brick-mapped PCG kernels (0: estimated from the last landed brick count) -- results do not depend on it;
 * "dense_tile_quads" 256|512|1024, "dense_tile_planes" n, "dense_grid" n: tile geometry / launch grid of the dense 2.5-D PCG kernels
 * (0 = default for the grid); "dense_kd_nt" -1|0|1: non-temporal stores of the dense direction kernel's output (-1: by grid size);
 * "dense_alternate_march" -1|0..3 (default -1 = by grid size): odd z-chunks march downwards in the dense direction kernel (bit 0) / update kernel (bit 1):
 *   the workgroups either side of a chunk interface then reach it together and share its planes through the L2; only the summation order of the dots changes.
 * "list_launch_grid" n: launch grid of the kernels that loop over a brick list (0 = estimated from the latest brick counts); results do not depend on it.
 * "fuse_divergence" 0|1|2 (default 1; 2 = also across blub_fluid_run_stage calls, for tests): inside blub_fluid_step the brick-mapped velocity solve forms div u in its init kernel (same bits; 0 = the
 *   separate divergence kernel, which blub_fluid_run_stage and the dense mapping always use).
 * "p2g_compact" -1|0|1 (default -1): 1 = the P2G gather compacts each tile's non-empty lists, 0 = one lane per list cell (same results bit for
 *   bit), -1 = chosen per step from the particles per FLUID brick.
 * "bricks_two_kernel_build" 0|1: build the brick lists with the two-kernel scan that grids with more 1024-brick blocks than CUs use.
 * "spin_free" 0|1: 1 = no kernel of blub_fluid_step waits for co-resident workgroups (two-kernel list build, no persistent PCG tail).  The default (0) uses two
 *   such kernels -- each with a bounded wait whose time-out is sticky, reported and survivable (BLUB_ERR_DEVICE, then this mode by itself) -- because they are
 *   worth 3 % of the metric (1 481-1 486 against 1 434-1 443 steps/s, profiles/r05_spin_free_cost.txt); a shared, partitioned or preemptible device should set it.
 * Unknown names: BLUB_ERR_INVALID_ARGUMENT. */
int blub_fluid_set_tuning(blub_fluid* h, const char* name, int value);
/* Upper bound on the steps the host may enqueue ahead of the GPU (default 4, further limited so that < ~700 kernel launches are queued; 0 = unbounded). blub_fluid_step blocks
 * (polling pinned memory) until step n - max has finished. */
int blub_fluid_set_max_steps_in_flight(blub_fluid* h, uint32_t max_steps);
/* {fluid bricks, active bricks, reset-list entries, stale bricks, total bricks, cells per brick} of the latest list build; blocks. */
int blub_fluid_get_brick_counts(blub_fluid* h, uint32_t out[6]);
/* Total PCG iterations executed (sum of reported iteration counts) since creation, both solvers. */
uint64_t blub_fluid_total_solver_iterations(const blub_fluid* h);

/* ---- step scheduler: `SimulationController` + `Timer` (src/simulation_controller.rs, src/timer.rs) ---------------
 * Host only, integer-nanosecond `Duration` arithmetic like the reference.  Stepping goes through callbacks so that a host can run
 * its whole `Scene::step` (animate models, voxelise, fluid step: scene/mod.rs:166-213); the *_fluid variants step a bare fluid. */
typedef struct blub_controller blub_controller;
typedef enum blub_controller_status {            /* SimulationControllerStatus, simulation_controller.rs:12-17 */
    BLUB_CONTROLLER_REALTIME = 0, BLUB_CONTROLLER_RECORDING = 1, BLUB_CONTROLLER_FAST_FORWARD = 2, BLUB_CONTROLLER_PAUSED = 3
} blub_controller_status;
typedef struct blub_step_callbacks {
    /* Scene::step for one simulation step: dt = Timer::simulation_delta().as_secs_f32(); total_simulated_time_ns already includes this
     * step (timer.rs:124).  Return BLUB_OK or an error (which ends the frame / fast-forward and is passed on). */
    int (*step)(void* user, float simulation_delta_seconds, uint64_t total_simulated_time_ns);
    int (*wait)(void* user);                     /* device.poll(Maintain::Wait), simulation_controller.rs:140; may be NULL */
    void* user;
} blub_step_callbacks;
int blub_controller_create(uint64_t simulation_steps_per_second /* 0 = the default 120 */, blub_controller** out);   /* ::new, :38-50 */
void blub_controller_destroy(blub_controller* c);
int blub_controller_set_simulation_steps_per_second(blub_controller* c, uint64_t steps_per_second);                  /* :88-92 */
uint64_t blub_controller_simulation_steps_per_second(const blub_controller* c);                                      /* :64 */
uint64_t blub_controller_simulation_delta_ns(const blub_controller* c);              /* Timer::simulation_delta: 1e9 / steps per second, :33-35 */
uint64_t blub_controller_total_simulated_time_ns(const blub_controller* c);          /* Timer::total_simulated_time */
uint64_t blub_controller_total_render_time_ns(const blub_controller* c);             /* Timer::total_render_time */
uint32_t blub_controller_num_simulation_steps_performed(const blub_controller* c);
uint32_t blub_controller_num_simulation_steps_performed_for_current_frame(const blub_controller* c);
uint64_t blub_controller_computation_time_last_fast_forward_ns(const blub_controller* c);                            /* :60, 147 */
int blub_controller_get_status(const blub_controller* c);                                                                /* :68 */
int blub_controller_set_simulation_stop_time_ns(blub_controller* c, uint64_t t);     /* pub simulation_stop_time (default one hour) */
uint64_t blub_controller_simulation_stop_time_ns(const blub_controller* c);
int blub_controller_set_time_scale(blub_controller* c, float time_scale);            /* pub time_scale */
int blub_controller_pause_or_resume(blub_controller* c);                                                             /* :72-78 */
int blub_controller_start_recording_with_fixed_frame_length(blub_controller* c, double frames_per_second);           /* :80-82 */
int blub_controller_restart(blub_controller* c);                                                                     /* :94-96 */
/* Timer::on_frame_submitted (timer.rs:76-88).  measured_frame_duration_ns < 0: measure the real time since the last call. */
int blub_controller_on_frame_submitted(blub_controller* c, int64_t measured_frame_duration_ns);                      /* :56-58 */
/* One rendered frame: step while the simulation lags the render clock; in real-time mode give up (accept lag) once the steps of
 * this frame cover more than 1/50 s of simulated time (:31, 159-217; timer.rs:94-130). */
int blub_controller_frame_steps(blub_controller* c, const blub_step_callbacks* cb, uint32_t* steps_out);
/* Batches of 16 steps + wait until max(jump, one step) of simulated time has passed (:96-157).  Like in the reference the status is
 * PAUSED afterwards (the stop-time mechanism ends the jump) and the wall clock of the whole jump is kept. */
int blub_controller_fast_forward_steps(blub_controller* c, uint64_t simulation_jump_length_ns, const blub_step_callbacks* cb, uint32_t* steps_out);
int blub_controller_frame_steps_fluid(blub_controller* c, blub_fluid* h, uint32_t* steps_out);
int blub_controller_fast_forward_steps_fluid(blub_controller* c, uint64_t simulation_jump_length_ns, blub_fluid* h, uint32_t* steps_out);

/* ---- z-slab domain decomposition over up to 8 GPUs (not part of the reference, which is single-GPU) ---- */
/* The global grid is cut into `num_slabs` z-ranges of whole 16x8x4-cell bricks.  Every slab keeps its volumes in GLOBAL
 * grid coordinates but only works on its own planes; neighbours exchange ghost particles, halo planes (RCCL
 * send/recv between z-neighbours), the PCG scalars (all-reduce) and migrating particles.  Two transports:
 *   create_local : all slabs in THIS process on one device (validation of the protocol on a single GPU),
 *   create_rccl  : one slab per process / GPU; `unique_id_128` comes from blub_rccl_unique_id() on rank 0 and is
 *                  distributed by the caller (e.g. torch.distributed broadcast). */
typedef struct blub_slab_group blub_slab_group;
int blub_rccl_unique_id(void* out128);
int blub_slab_range(uint32_t nz, int num_slabs, int index, int32_t* z0, int32_t* z1);   /* host only */
int blub_slab_group_create_local(const blub_fluid_desc* desc, int num_slabs, blub_slab_group** out);
int blub_slab_group_create_rccl(const blub_fluid_desc* desc, int rank, int num_ranks, const void* unique_id_128, blub_slab_group** out);
/* The same with the caller's cut planes instead of uniform ones and a choice of the memory the slabs live in.
 * cuts[num_slabs + 1]: cuts[0] = 0, strictly increasing multiples of the brick depth (4), cuts[num_slabs] = nz; slab r owns the planes [cuts[r], cuts[r + 1]).
 *   NULL = uniform.  Every slab allocates the plane count of the thickest one (one export layout for the direct transport).  Every rank passes the same
 *   cuts.  (Round-4 review: the metric's scene keeps its fluid in z < 32 and z >= 224 of 256 -- uniform cuts into 8 leave six ranks without fluid.)
 * memory_mode (what the DIRECT transport's peers write into while kernels run: the volume slab with its ghost planes, the three recurrence volumes, the
 *   exchange arena with flags / partials / particle staging): BLUB_SLAB_MEMORY_COARSE = hipMalloc.  HIP promises visibility of another agent's writes
 *   to coarse-grained memory only at kernel boundaries; the transport's write-through stores and cache-bypassing loads (sc0 sc1) are measured to work
 *   between the XCDs of one device and between processes on one device, but across xGMI that is outside the documented model.
 *   BLUB_SLAB_MEMORY_FINE_GRAINED / _UNCACHED = hipExtMallocWithFlags(hipDeviceMallocFinegrained / hipDeviceMallocUncached): the documented way to share
 *   memory between agents while kernels run; cost on one device: profiles/r05_slab_memory_modes.jsonl.  bench.py's probe (blub_amd/direct_probe.py)
 *   tries COARSE first, then FINE_GRAINED, then falls back to RCCL.  Every rank passes the same mode. */
enum { BLUB_SLAB_MEMORY_COARSE = 0, BLUB_SLAB_MEMORY_FINE_GRAINED = 1, BLUB_SLAB_MEMORY_UNCACHED = 2,
       /* OR-ed into memory_mode: every slab allocates the WHOLE grid (1.2 GiB at 256^3, 8.5 GiB at 512^3 -- of 288) instead of its own planes + 8 on either
        * side, so that the cut planes can move while the group runs (blub_slab_group_recut / _rebalance) */
       BLUB_SLAB_FULL_VOLUMES = 0x100 };
int blub_slab_group_create_local_ex(const blub_fluid_desc* desc, int num_slabs, const int32_t* cuts, uint32_t memory_mode, blub_slab_group** out);
int blub_slab_group_create_rccl_ex(const blub_fluid_desc* desc, int rank, int num_ranks, const void* unique_id_128, const int32_t* cuts, uint32_t memory_mode, blub_slab_group** out);
/* Host only: cut planes that give every slab about the same number of FLUID bricks for the given particle positions (16-byte records, as
 * blub_fluid_set_particles takes them) -- the contiguous partition of the brick layers that minimises the heaviest slab, every slab at least
 * `min_layers` brick layers thick (>= 1).  cuts_out: num_slabs + 1 planes; fluid_bricks_out (may be NULL): FLUID bricks per slab at these positions. */
int blub_slab_balanced_cuts(const uint32_t grid_dim[3], uint32_t num_particles, const float* pos_ll, int num_slabs, int min_layers, int32_t* cuts_out, uint32_t* fluid_bricks_out);
/* ---- checkpoints and in-place recovery of a group on the DIRECT transport (round 5) ----
 * A bounded wait for a peer that runs out (a peer seconds late, or gone) used to end the group: every later wait gives up at once, the data stepped since is
 * invalid, and the caller had to tear the job down.  With an interval > 0 every local slab copies its restartable state -- particles, the two pressure
 * volumes, the counts (everything else is scratch) -- into one of two device-side generations at the start of every `every_n_steps`-th
 * step, and only while its time-out mark is clear: every generation a rank holds predates its first failed wait.  Recovery, by the caller over its own
 * control plane (blub_amd.SlabGroup.recover_over_torch_distributed is the worked example; tests/test_gpu_multirank.py injects a 10-second stall):
 *   1. every rank: blub_slab_group_synchronize (drains the stream; reports BLUB_ERR_COMM once and clears the mark);
 *   2. all-gather blub_slab_group_checkpoints and blub_slab_group_exchange_sequence; pick the newest step EVERY rank holds -- a generation taken after
 *      some rank's failure is missing on that rank -- and a sequence base above every rank's number;  3. barrier;
 *   4. every rank: blub_slab_group_restore(step, base);  5. barrier; step on (replaying from `step`).
 * Cost of a generation: one pass over 64 B per particle + 8 B per held cell (~40 us for the metric's scene), amortised over the interval. */
int blub_slab_group_set_checkpoint_interval(blub_slab_group* g, uint32_t every_n_steps);      /* 0 = off (default); every rank the same value.  Between PROCESSES use >= 4: a rank that was only late keeps writing generations for a step or two before it sees a mark, and with 1 - 2 it could replace both generations its peers still hold in common */
int blub_slab_group_checkpoints(blub_slab_group* g, uint32_t steps_out[2]);                    /* step numbers of the two generations, 0xFFFFFFFF = none; blocks */
int blub_slab_group_exchange_sequence(const blub_slab_group* g, uint32_t* seq_out);
int blub_slab_group_restore(blub_slab_group* g, uint32_t step, uint32_t sequence_base);
/* ---- moving the cut planes of a running group (round 5) ----
 * Cuts balanced for the particles at t = 0 go stale: the metric's dam break spreads over all of z within ~100 steps (profiles/r05_slab_cuts_uniform_vs_weighted.jsonl).
 * blub_slab_group_recut: COLLECTIVE, between steps, groups created with BLUB_SLAB_FULL_VOLUMES.  Every new cut plane must lie strictly between its two old
 *   neighbours -- state then only moves between ADJACENT slabs: the planes of the two pressure volumes that change owner travel like halo planes, the
 *   particles through one ordinary migration exchange against the new ranges; everything else is scratch.  One host synchronisation.
 * blub_slab_group_rebalance: COLLECTIVE.  FLUID bricks per brick layer are counted on the device and gathered, every rank derives the same balanced cuts
 *   (the partition of blub_slab_balanced_cuts, every slab >= min_layers brick layers), clamped to what ONE re-cut may move, and the group re-cuts if that
 *   lowers the heaviest slab's share of the FLUID bricks by more than 5 %.  *changed (may be NULL) = 1 if the cuts moved.  Call it every few dozen steps. */
int blub_slab_group_recut(blub_slab_group* g, const int32_t* new_cuts);
int blub_slab_group_rebalance(blub_slab_group* g, int min_layers, int* changed);
/* The cut planes of a group (num_slabs + 1 values, the last one = nz). */
int blub_slab_group_cuts(const blub_slab_group* g, int32_t* cuts_out);
void blub_slab_group_destroy(blub_slab_group* g);
int blub_slab_group_num_local(const blub_slab_group* g);
/* For read_volume / statistics of one slab (borrowed: do not destroy).  A slab holds the planes [z0 - 8, z1 + 8) of every grid volume only (its own
 * range plus two brick layers: ghost particles, the dilated active bricks, stale bricks): blub_fluid_read_volume fills the rest of the
 * full-grid host array with zeros, blub_fluid_write_volume / set_solid_voxels take the held planes of the full-grid input. */
blub_fluid* blub_slab_group_local_fluid(blub_slab_group* g, int local_index);
int blub_slab_group_local_range(const blub_slab_group* g, int local_index, int32_t* z0, int32_t* z1);
/* every rank passes the same global arrays; each slab keeps the particles of its z-range */
int blub_slab_group_set_particles(blub_slab_group* g, uint32_t n, const float* pos_ll, const float* vx, const float* vy, const float* vz);
uint32_t blub_slab_group_num_particles(blub_slab_group* g);   /* blocks: the counts live on the device between steps */   /* own particles of the local slabs */
int blub_slab_group_get_particles(blub_slab_group* g, float* pos_ll, float* vx, float* vy, float* vz);
int blub_slab_group_set_gravity_grid(blub_slab_group* g, const float gravity_grid[3]);
int blub_slab_group_set_solver_config(blub_slab_group* g, int which, const blub_solver_config* cfg);
int blub_slab_group_set_rebinning_frequency(blub_slab_group* g, uint32_t every_n_steps);
int blub_slab_group_set_pcg_schedule(blub_slab_group* g, int mode);   /* blub_fluid_set_pcg_schedule on every local slab; all ranks must agree */
/* RCCL transport of the PCG partials: 0 = grouped send/recv fused with the halo planes, 1 = ncclAllGather next to a p2p-only halo group.
 * Calibrated at creation (both are timed on the hardware at hand); this overrides the choice.  All ranks must pass the same mode. */
int blub_slab_group_set_gather_mode(blub_slab_group* g, int mode);
/* static objects (see blub_fluid_set_meshes / blub_fluid_voxelize): every local slab voxelises the meshes in global grid coordinates */
int blub_slab_group_set_meshes(blub_slab_group* g, uint32_t num_vertices, const float* positions_xyz, uint32_t num_indices, const uint32_t* indices);
int blub_slab_group_voxelize(blub_slab_group* g, uint32_t num_meshes, const blub_mesh_desc* meshes);
/* Particle exchanges without host synchronisation (default on): from the second step after the particles were set, ghost copies and
 * migrating particles travel in fixed-capacity messages with a count header (capacity = 1.5 x the count of the same exchange one step
 * earlier + 2048; an overflow makes the NEXT step return BLUB_ERR_OUT_OF_MEMORY) and the particle counts stay on the device.  0: the
 * synchronous protocol of rounds 1-2 (counts first, one stream synchronisation per exchange).  All ranks must pass the same value. */
int blub_slab_group_set_async_exchange(blub_slab_group* g, int enabled);
/* diagnostics: stream synchronisations issued inside blub_slab_group_step so far -- by particle exchanges / by looks at a solve's `done` */
int blub_slab_group_host_syncs(const blub_slab_group* g, uint64_t* particle_exchanges, uint64_t* done_polls);
/* Particles a migration HELD BACK at their sender for one exchange, plus ghost copies left out for one step, because a message of the
 * host-synchronisation-free exchange was sized too small from the previous step's count (a front reaching an interface that carried
 * nothing before).  Nothing is lost: a held-back particle stays an own particle of its slab, clamped just inside the range, and travels
 * with the next exchange (whose message is sized for it).  Sum over the local slabs since creation; blocks. */
uint64_t blub_slab_group_held_back(blub_slab_group* g);
/* Transport of a z-slab group.  0: host-issued operations between the kernels (device copies inside a local group; grouped RCCL send / recv
 * between processes) -- the default of blub_slab_group_create_rccl.  1: DIRECT -- every slab stores what its neighbours need straight into
 * THEIR memory (the same process in a local group, hipIpc mappings between processes: xGMI peer access on one node) and raises a flag word;
 * consumers wait for the flags on the device (inside the PCG iteration kernel; a one-block wait kernel elsewhere).  No host-issued transport
 * operation, no stream synchronisation, no message sizes (a particle exchange cannot outgrow anything but the particle capacity itself).
 * The default of blub_slab_group_create_local.  Between steps only; every rank passes the same value; with several processes every peer
 * must have been connected (below).  Results are the same to the bit as with transport 0 (same kernels, same order of every sum). */
int blub_slab_group_set_transport(blub_slab_group* g, int kind);
int blub_slab_group_get_transport(const blub_slab_group* g);
/* DIRECT transport between processes: export() fills `out` (export_size() bytes: one hipIpc handle + size per exportable allocation of the
 * local slab) -- carry it to the other ranks by any means (bench.py: torch.distributed.all_gather_object) and hand rank r's blob to connect(). */
int blub_slab_group_export_size(const blub_slab_group* g);
int blub_slab_group_export(blub_slab_group* g, void* out, int capacity);
int blub_slab_group_connect(blub_slab_group* g, int rank, const void* blob, int bytes);
/* TEST HOOK: runs segments [first, last] of ONE step so that a test can look at every slab in between: 0 ghost-particle exchange,
 * 1 transfer, 2 divergence, 3 solve_velocity, 4 binning, 5 project (+ extrapolation), 6 advect, 7 migration + density ghosts,
 * 8 density_gather, 9 solve_density, 10 position_change (+ extrapolation), 11 correct, 12 second migration, 13 step counter.
 * blub_slab_group_step == run_stages(0, 13).  All ranks must pass the same range. */
int blub_slab_group_run_stages(blub_slab_group* g, float simulation_delta, int first, int last);
int blub_slab_group_step(blub_slab_group* g, float simulation_delta_seconds);
int blub_slab_group_synchronize(blub_slab_group* g);
/* diagnostics: grouped transport operations (halo / partial / particle exchanges) issued by this process so far */
uint64_t blub_slab_group_transport_ops(const blub_slab_group* g);
/* "loopback" or the RCCL transport with the result of its start-up calibration (valid until the group is destroyed) */
const char* blub_slab_group_transport_description(const blub_slab_group* g);

#ifdef __cplusplus
}
#endif
#endif /* BLUBHIP_H */
