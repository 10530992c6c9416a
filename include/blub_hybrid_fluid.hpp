// blub_hybrid_fluid.hpp -- the host side above the C-ABI (blubhip.h) in a compiled language: C++17, header only.
//
// The reference's host code is Rust (src/simulation/hybrid_fluid.rs, src/scene/mod.rs, src/simulation_controller.rs); this image has no Rust
// toolchain, so the shim a maintainer would write in Rust (the generated `extern "C"` block + `impl HybridFluid`) exists here as the
// same surface in C++ -- the same type and method names, argument meaning and error behaviour, one method per reference method, each citing the line
// it stands for.  Nothing in here computes: every method is one or two calls through the C-ABI into libblubhip.so.  There is no CPU path: without a HIP
// device `HybridFluid`'s constructor throws blub::Error{BLUB_ERR_NO_DEVICE}.
//
// Differences a port has to know (all forced by "no wgpu"):
//   * `device` / `queue` / `shader_dir` / `pipeline_manager` / `per_frame_bind_group_layout` arguments do not exist; the HIP device ordinal takes their place.
//   * `step` takes no encoder: it enqueues on the engine's own stream (bind_group_renderer().stream) and returns; `Scene::step` does not submit.
//   * `bind_group_renderer()` returns device pointers (blub_device_views) instead of a wgpu::BindGroup.
//   * `&mut SolverConfig` accessors: C++ cannot hand out a reference into the engine, so pressure_solver_config_*() returns a proxy that writes back
//     when it goes out of scope (`fluid.pressure_solver_config_velocity()->max_num_iterations = 64;`).
//   * errors the reference `panic!`s / `unwrap()`s on are blub::Error exceptions carrying the blub_status and blub_last_error_string().
#ifndef BLUB_HYBRID_FLUID_HPP
#define BLUB_HYBRID_FLUID_HPP

#include <array>
#include <chrono>
#include <cstdint>
#include <deque>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "blubhip.h"

namespace blub {

using Duration = std::chrono::nanoseconds;      // std::time::Duration: integer nanoseconds
using Point3 = std::array<float, 3>;            // cgmath::Point3<f32>
using Vector3 = std::array<float, 3>;           // cgmath::Vector3<f32>
struct Extent3d { uint32_t width, height, depth; };      // wgpu::Extent3d

struct Error : std::runtime_error {
    int status;
    Error(int status_, const std::string& what) : std::runtime_error(what), status(status_) {}
};
inline void check(int rc) {
    if (rc != BLUB_OK) throw Error(rc, std::string(blub_last_error_string()));
}
// Duration::as_secs_f32 (what HybridFluid::step feeds the shaders, hybrid_fluid.rs:778)
inline float as_secs_f32(Duration d) {
    const int64_t ns = d.count();
    return (float)(ns / 1000000000) + (float)(ns % 1000000000) / 1e9f;      // (secs as f32) + (nanos as f32) / 1e9: core::time, as the scheduler does it
}

struct SolverConfig {                           // pressure_solver.rs:57-62
    float error_tolerance = 0.1f;
    int32_t error_check_frequency = 4;
    int32_t max_num_iterations = 32;
};
struct SolverStatisticSample {                  // pressure_solver.rs:64-68
    float error;
    int32_t iteration_count;
};
struct DynamicSettings {                        // hybrid_fluid.rs:19-22
    uint32_t particle_rebinning_step_frequency = 60;
};

class HybridFluid {
  public:
    static constexpr uint32_t PARTICLES_PER_GRID_CELL = BLUB_PARTICLES_PER_GRID_CELL;      // hybrid_fluid.rs:90

    // HybridFluid::new (hybrid_fluid.rs:92-607)
    HybridFluid(Extent3d grid_dimension, uint32_t max_num_particles, int32_t device = -1) {
        blub_fluid_desc d{};
        d.nx = grid_dimension.width; d.ny = grid_dimension.height; d.nz = grid_dimension.depth;
        d.max_num_particles = max_num_particles;
        d.device = device;
        check(blub_fluid_create(&d, &h_));
    }
    // Scene::create_fluid_from_config (scene/mod.rs:109-144)
    HybridFluid(const blub_scene_config& scene, int32_t device = -1) { check(blub_fluid_create_from_scene(&scene, device, &h_)); }
    ~HybridFluid() { if (h_) blub_fluid_destroy(h_); }
    HybridFluid(HybridFluid&& o) noexcept : h_(std::exchange(o.h_, nullptr)), stats_(std::move(o.stats_)) {}
    HybridFluid& operator=(HybridFluid&& o) noexcept {
        if (this != &o) { if (h_) blub_fluid_destroy(h_); h_ = std::exchange(o.h_, nullptr); stats_ = std::move(o.stats_); }
        return *this;
    }
    HybridFluid(const HybridFluid&) = delete;
    HybridFluid& operator=(const HybridFluid&) = delete;

    // :620-678.  Adds 8 jittered particles per cell of the box (grid space); truncates at max_num_particles like the reference (which logs `error!`):
    // last_add_dropped() says how many did not fit.
    void add_fluid_cube(Point3 min_grid, Point3 max_grid) { check(blub_fluid_add_fluid_cube(h_, min_grid.data(), max_grid.data())); }
    uint32_t last_add_dropped() const { return blub_fluid_last_add_dropped(h_); }
    // :680-690 update_signed_distance_field_for_static is an empty stub in the reference (the solids reach the fluid through the voxelisation volume): so is this.
    void update_signed_distance_field_for_static() {}
    void set_gravity_grid(Vector3 gravity) { check(blub_fluid_set_gravity_grid(h_, gravity.data())); }      // :692
    uint32_t num_particles() const { return blub_fluid_num_particles(h_); }                                 // :696
    uint32_t num_active_particles() const { return blub_fluid_num_particles(h_); }                          // :731 (the reference keeps one count)
    uint32_t max_num_particles() const { return blub_fluid_max_num_particles(h_); }
    Extent3d grid_dimension() const {                                                                        // :727
        uint32_t d[3];
        check(blub_fluid_grid_dimension(h_, d));
        return Extent3d{d[0], d[1], d[2]};
    }
    // :700-725 -- read-only access for a renderer.  Re-query after every step (include/blubhip.h: the particle buffers swap).
    blub_device_views bind_group_renderer() const {
        blub_device_views v{};
        check(blub_fluid_get_device_views(h_, &v));
        return v;
    }

    // :743-749 `&mut SolverConfig`
    class SolverConfigRef {
      public:
        SolverConfigRef(blub_fluid* h, int which) : h_(h), which_(which) {
            blub_solver_config c{};
            check(blub_fluid_get_solver_config(h_, which_, &c));
            cfg_.error_tolerance = c.error_tolerance; cfg_.max_num_iterations = c.max_num_iterations; cfg_.error_check_frequency = c.error_check_frequency;
        }
        ~SolverConfigRef() noexcept(false) {
            blub_solver_config c{cfg_.error_tolerance, cfg_.max_num_iterations, cfg_.error_check_frequency};
            const int rc = blub_fluid_set_solver_config(h_, which_, &c);
            if (rc != BLUB_OK && !std::uncaught_exceptions()) throw Error(rc, blub_last_error_string());
        }
        SolverConfig* operator->() { return &cfg_; }
        SolverConfig& operator*() { return cfg_; }
      private:
        blub_fluid* h_; int which_; SolverConfig cfg_;
    };
    SolverConfigRef pressure_solver_config_velocity() { return SolverConfigRef(h_, BLUB_SOLVER_VELOCITY); }      // :743
    SolverConfigRef pressure_solver_config_density() { return SolverConfigRef(h_, BLUB_SOLVER_DENSITY); }        // :747

    // :751 `&mut DynamicSettings`
    class DynamicSettingsRef {
      public:
        explicit DynamicSettingsRef(blub_fluid* h) : h_(h) { s_.particle_rebinning_step_frequency = blub_fluid_get_rebinning_frequency(h_); }
        ~DynamicSettingsRef() noexcept(false) {
            const int rc = blub_fluid_set_rebinning_frequency(h_, s_.particle_rebinning_step_frequency);
            if (rc != BLUB_OK && !std::uncaught_exceptions()) throw Error(rc, blub_last_error_string());
        }
        DynamicSettings* operator->() { return &s_; }
      private:
        blub_fluid* h_; DynamicSettings s_;
    };
    DynamicSettingsRef dynamic_settings() { return DynamicSettingsRef(h_); }

    // :755-761: the history of <= 100 samples (pressure_solver.rs:101), oldest first
    const std::deque<SolverStatisticSample>& pressure_solver_stats_velocity() const { return refresh(BLUB_SOLVER_VELOCITY); }
    const std::deque<SolverStatisticSample>& pressure_solver_stats_density() const { return refresh(BLUB_SOLVER_DENSITY); }
    void update_statistics() { check(blub_fluid_update_statistics(h_)); }                                   // :765, non-blocking

    // :770-977.  Enqueues one simulation step; returns before the GPU is done.
    void step(Duration simulation_delta) { check(blub_fluid_step(h_, as_secs_f32(simulation_delta))); }
    // device.poll(Maintain::Wait) (scene/mod.rs:142, simulation_controller.rs:140)
    void synchronize() { check(blub_fluid_synchronize(h_)); }

    // ---- beyond the reference's surface: what its wgpu buffers give a host for free ----
    // particles_position_llindex / particles_velocity_{x,y,z} as host arrays of float4, in the caller's order; any pointer may be null
    void set_particles(uint32_t n, const float* pos_ll, const float* vx, const float* vy, const float* vz) { check(blub_fluid_set_particles(h_, n, pos_ll, vx, vy, vz)); }
    void get_particles(float* pos_ll, float* vx, float* vy, float* vz) { check(blub_fluid_get_particles(h_, pos_ll, vx, vy, vz)); }
    std::vector<float> particle_positions() {
        std::vector<float> p(4 * (size_t)num_particles());
        if (!p.empty()) get_particles(p.data(), nullptr, nullptr, nullptr);
        return p;
    }
    void read_volume(blub_volume which, void* host_out) { check(blub_fluid_read_volume(h_, which, host_out)); }
    // the voxelisation volume the reference borrows from SceneVoxelization (hybrid_fluid.rs:266)
    void set_meshes(const std::vector<float>& positions_xyz, const std::vector<uint32_t>& indices) {
        check(blub_fluid_set_meshes(h_, (uint32_t)(positions_xyz.size() / 3), positions_xyz.data(), (uint32_t)indices.size(), indices.data()));
    }
    void voxelize(const std::vector<blub_mesh_desc>& meshes) { check(blub_fluid_voxelize(h_, (uint32_t)meshes.size(), meshes.data())); }
    uint64_t total_solver_iterations() const { return blub_fluid_total_solver_iterations(h_); }
    // performance choices the engine otherwise makes from its own counters (include/blubhip.h: blub_fluid_set_tuning)
    void set_tuning(const std::string& name, int value) { check(blub_fluid_set_tuning(h_, name.c_str(), value)); }
    blub_fluid* handle() { return h_; }

  private:
    const std::deque<SolverStatisticSample>& refresh(int which) const {
        std::deque<SolverStatisticSample>& q = stats_[which];
        q.clear();
        const int n = blub_fluid_solver_stats_count(h_, which);
        for (int i = 0; i < n; ++i) {
            blub_solver_stats s{};
            check(blub_fluid_solver_stats_get(h_, which, i, &s));
            q.push_back(SolverStatisticSample{s.error, s.iteration_count});
        }
        return q;
    }
    blub_fluid* h_ = nullptr;
    mutable std::array<std::deque<SolverStatisticSample>, 2> stats_;
};

// src/scene/mod.rs: the scene file, the fluid made from it, the static objects voxelised before every step
class Scene {
  public:
    // Scene::new (scene/mod.rs:56-99).  models_dir: where the `model` paths of static objects are relative to (models.rs:267)
    explicit Scene(const std::string& scene_path, int32_t device = -1, std::string models_dir = std::string()) : device_(device), models_dir_(std::move(models_dir)) {
        check(blub_scene_load_json(scene_path.c_str(), &config_));
        if (models_dir_.empty()) {      // the `models` directory next to the scene file
            const size_t slash = scene_path.find_last_of('/');
            models_dir_ = (slash == std::string::npos ? std::string(".") : scene_path.substr(0, slash)) + "/models";
        }
        reset();
    }
    const blub_scene_config& config() const { return config_; }                         // :101
    uint32_t num_active_particles() const { return fluid_->num_active_particles(); }     // :105
    // :146-164
    void reset() {
        fluid_.reset();
        fluid_ = std::make_unique<HybridFluid>(config_, device_);
        meshes_.clear();
        models_loaded_ = false;
        total_simulated_time_ = Duration(0);
    }
    // :166-213: animate the models, voxelise them, step the fluid, poll the statistics.  The timer has already advanced by the step being taken when
    // Scene::step runs (timer.rs:124): total_simulated_time includes it.
    void step(Duration simulation_delta) { step(simulation_delta, total_simulated_time_ + simulation_delta); }
    // ... with the clock of the caller's Timer, like the reference (`timer.total_simulated_time()`: the scene has no clock of its own there)
    void step(Duration simulation_delta, Duration total_simulated_time) {
        total_simulated_time_ = total_simulated_time;
        if (config_.num_static_objects) {
            if (!models_loaded_) load_models();
            std::vector<blub_mesh_desc> descs;
            for (const Mesh& m : meshes_) {
                blub_mesh_desc d{};
                check(blub_scene_mesh_desc_at_time(&config_, m.object, (uint64_t)total_simulated_time_.count(), (uint64_t)simulation_delta.count(), &d));
                d.index_begin = m.begin; d.index_end = m.end;
                descs.push_back(d);
            }
            fluid_->voxelize(descs);
        }
        fluid_->step(simulation_delta);
        fluid_->update_statistics();
    }
    const HybridFluid& fluid() const { return *fluid_; }      // :216
    HybridFluid& fluid_mut() { return *fluid_; }              // :220
    Duration total_simulated_time() const { return total_simulated_time_; }

  private:
    struct Mesh { uint32_t object, begin, end; };
    // SceneModels::from_config (scene/models.rs:255-376): one shared vertex / index buffer, one index range per object
    void load_models() {
        std::vector<float> positions;
        std::vector<uint32_t> indices;
        for (uint32_t i = 0; i < config_.num_static_objects; ++i) {
            const std::string path = models_dir_ + "/" + config_.static_objects[i].model;
            uint32_t nv = 0, ni = 0;
            check(blub_load_obj(path.c_str(), nullptr, 0, &nv, nullptr, 0, &ni));
            std::vector<float> p(3 * (size_t)nv);
            std::vector<uint32_t> idx(ni);
            check(blub_load_obj(path.c_str(), p.data(), nv, &nv, idx.data(), ni, &ni));
            const uint32_t v0 = (uint32_t)(positions.size() / 3), i0 = (uint32_t)indices.size();
            positions.insert(positions.end(), p.begin(), p.end());
            for (uint32_t k : idx) indices.push_back(k + v0);
            meshes_.push_back(Mesh{i, i0, i0 + ni});
        }
        if (!meshes_.empty()) fluid_->set_meshes(positions, indices);
        models_loaded_ = true;
    }
    blub_scene_config config_{};
    int32_t device_;
    std::string models_dir_;
    std::unique_ptr<HybridFluid> fluid_;
    std::vector<Mesh> meshes_;
    bool models_loaded_ = false;
    Duration total_simulated_time_{0};
};

enum class SimulationControllerStatus { Realtime = BLUB_CONTROLLER_REALTIME, RecordingWithFixedFrameLength = BLUB_CONTROLLER_RECORDING,
                                        FastForward = BLUB_CONTROLLER_FAST_FORWARD, Paused = BLUB_CONTROLLER_PAUSED };      // simulation_controller.rs:12-17

// src/simulation_controller.rs + src/timer.rs: when to step, how often, for how long
class SimulationController {
  public:
    SimulationController() { check(blub_controller_create(0, &c_)); }                    // ::new, :38-50 (120 steps per second)
    ~SimulationController() { if (c_) blub_controller_destroy(c_); }
    SimulationController(const SimulationController&) = delete;
    SimulationController& operator=(const SimulationController&) = delete;

    void on_frame_submitted() { check(blub_controller_on_frame_submitted(c_, -1)); }      // :55
    void on_frame_submitted(Duration measured) { check(blub_controller_on_frame_submitted(c_, measured.count())); }
    Duration computation_time_last_fast_forward() const { return Duration((int64_t)blub_controller_computation_time_last_fast_forward_ns(c_)); }      // :59
    uint64_t simulation_steps_per_second() const { return blub_controller_simulation_steps_per_second(c_); }      // :63
    SimulationControllerStatus status() const { return (SimulationControllerStatus)blub_controller_get_status(c_); }      // :67
    void pause_or_resume() { check(blub_controller_pause_or_resume(c_)); }                // :71
    void start_recording_with_fixed_frame_length(double frames_per_second) { check(blub_controller_start_recording_with_fixed_frame_length(c_, frames_per_second)); }      // :79
    void set_simulation_steps_per_second(uint64_t n) { check(blub_controller_set_simulation_steps_per_second(c_, n)); }      // :83
    void restart() { check(blub_controller_restart(c_)); }                                // :89
    void set_simulation_stop_time(Duration t) { check(blub_controller_set_simulation_stop_time_ns(c_, (uint64_t)t.count())); }      // pub simulation_stop_time
    void set_time_scale(float s) { check(blub_controller_set_time_scale(c_, s)); }        // pub time_scale
    // Timer (timer.rs:128-162)
    Duration simulation_delta() const { return Duration((int64_t)blub_controller_simulation_delta_ns(c_)); }
    Duration total_simulated_time() const { return Duration((int64_t)blub_controller_total_simulated_time_ns(c_)); }
    Duration total_render_time() const { return Duration((int64_t)blub_controller_total_render_time_ns(c_)); }
    uint32_t num_simulation_steps_performed() const { return blub_controller_num_simulation_steps_performed(c_); }
    uint32_t num_simulation_steps_performed_for_current_frame() const { return blub_controller_num_simulation_steps_performed_for_current_frame(c_); }

    // :96-157.  Returns the number of steps taken; the status is Paused afterwards, like in the reference.
    uint32_t fast_forward_steps(Duration simulation_jump_length, Scene& scene) {
        uint32_t steps = 0;
        const blub_step_callbacks cb = callbacks(scene);
        check(blub_controller_fast_forward_steps(c_, (uint64_t)simulation_jump_length.count(), &cb, &steps));
        return steps;
    }
    // :159-217.  One rendered frame.
    uint32_t frame_steps(Scene& scene) {
        uint32_t steps = 0;
        const blub_step_callbacks cb = callbacks(scene);
        check(blub_controller_frame_steps(c_, &cb, &steps));
        return steps;
    }

    // the same two for a bare fluid (no static objects to animate): blub_fluid_step + update_statistics per step, synchronize per batch -- no callback in the loop
    uint32_t fast_forward_steps(Duration simulation_jump_length, HybridFluid& fluid) {
        uint32_t steps = 0;
        check(blub_controller_fast_forward_steps_fluid(c_, (uint64_t)simulation_jump_length.count(), fluid.handle(), &steps));
        return steps;
    }
    uint32_t frame_steps(HybridFluid& fluid) {
        uint32_t steps = 0;
        check(blub_controller_frame_steps_fluid(c_, fluid.handle(), &steps));
        return steps;
    }

  private:
    struct StepContext { Scene* scene; Duration simulation_delta; };
    blub_step_callbacks callbacks(Scene& scene) {
        ctx_ = StepContext{&scene, simulation_delta()};
        blub_step_callbacks cb{};
        cb.user = &ctx_;
        cb.step = [](void* user, float, uint64_t total_ns) -> int {
            StepContext& c = *static_cast<StepContext*>(user);
            try {
                c.scene->step(c.simulation_delta, Duration((int64_t)total_ns));      // Scene::step(&timer, ...): the timer has advanced by the step being taken
            } catch (const Error& e) { return e.status; }
            return BLUB_OK;
        };
        cb.wait = [](void* user) -> int {
            try { static_cast<StepContext*>(user)->scene->fluid_mut().synchronize(); } catch (const Error& e) { return e.status; }
            return BLUB_OK;
        };
        return cb;
    }
    StepContext ctx_{nullptr, Duration(0)};
    blub_controller* c_ = nullptr;
};

}  // namespace blub
#endif
