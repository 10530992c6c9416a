// A host of the engine in a compiled language: the reference's main loop (main.rs: Scene::new, SimulationController::fast_forward_steps / frame_steps,
// HybridFluid accessors) written against include/blub_hybrid_fluid.hpp -- no Python, no torch.  Built and run by tests/test_native_host.py.
//   hybrid_fluid_host --host-only <scene.json>                 no GPU needed: scene parsing, the cube generator, and the NO_DEVICE error of the constructor
//   hybrid_fluid_host --frames <frames> <positions.bin>        a fluid built by hand (HybridFluid::new, add_fluid_cube x 2, set_gravity_grid) driven frame by frame
//   hybrid_fluid_host <scene.json> <steps> <positions.bin> [tuning=value ...]     fast-forwards `steps` simulation steps with solves of a fixed 120 iterations, writes the particle positions
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "blub_hybrid_fluid.hpp"

static int host_only(const char* scene_path) {
    blub_scene_config cfg{};
    blub::check(blub_scene_load_json(scene_path, &cfg));
    // HybridFluid::add_fluid_cube's generator on the host (hybrid_fluid.rs:609-678): 8 particles per cell of the first cube, in grid space
    const float s = cfg.grid_to_world_scale;
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = (cfg.cube_min[0][k] - cfg.world_position[k]) / s; hi[k] = (cfg.cube_max[0][k] - cfg.world_position[k]) / s; }
    std::vector<float> pos(4 * (size_t)cfg.max_num_particles);
    uint32_t n = 0;
    blub::check(blub_seed_fluid_cube(cfg.grid_dimension, cfg.max_num_particles, 0, lo, hi, pos.data(), cfg.max_num_particles, &n));
    int status = BLUB_OK;
    std::string what;
    try {
        blub::HybridFluid fluid(blub::Extent3d{cfg.grid_dimension[0], cfg.grid_dimension[1], cfg.grid_dimension[2]}, cfg.max_num_particles);
        (void)fluid;
    } catch (const blub::Error& e) { status = e.status; what = e.what(); }
    int bad_status = BLUB_OK;
    try { blub::Scene missing("/nonexistent/scene.json"); } catch (const blub::Error& e) { bad_status = e.status; }
    std::printf("{\"grid\": [%u, %u, %u], \"cubes\": %u, \"seeded\": %u, \"create_status\": %d, \"missing_scene_status\": %d, \"version\": \"%s\"}\n", cfg.grid_dimension[0],
                cfg.grid_dimension[1], cfg.grid_dimension[2], cfg.num_fluid_cubes, n, status, bad_status, blub_version_string());
    return 0;
}

// main.rs's render loop without a renderer: on_frame_submitted (a fixed 1/60 s per frame, as start_recording_with_fixed_frame_length would give), frame_steps
static int frames(int num_frames, const char* out_path) {
    blub::HybridFluid fluid(blub::Extent3d{64, 48, 32}, 120000);
    fluid.add_fluid_cube({2.0f, 2.0f, 2.0f}, {30.0f, 34.0f, 30.0f});
    const uint32_t after_first = fluid.num_particles();
    fluid.add_fluid_cube({40.0f, 2.0f, 4.0f}, {62.0f, 20.0f, 28.0f});      // (truncated at max_num_particles, like the reference)
    fluid.set_gravity_grid({0.0f, -981.0f, 0.0f});
    for (int which = 0; which < 2; ++which) {
        auto cfg = which ? fluid.pressure_solver_config_density() : fluid.pressure_solver_config_velocity();
        cfg->error_tolerance = 0.0f; cfg->max_num_iterations = 120; cfg->error_check_frequency = 8;
    }
    blub::SimulationController controller;
    uint32_t steps = 0;
    std::string per_frame;
    for (int f = 0; f < num_frames; ++f) {
        controller.on_frame_submitted(blub::Duration(16666667));
        const uint32_t n = controller.frame_steps(fluid);
        steps += n;
        per_frame += (f ? ", " : "") + std::to_string(n);
    }
    fluid.synchronize();
    fluid.update_statistics();
    const std::vector<float> pos = fluid.particle_positions();
    if (FILE* f = std::fopen(out_path, "wb")) { std::fwrite(pos.data(), sizeof(float), pos.size(), f); std::fclose(f); }
    else { std::fprintf(stderr, "cannot write %s\n", out_path); return 1; }
    std::printf("{\"steps_taken\": %u, \"steps_per_frame\": [%s], \"num_particles\": %u, \"after_first_cube\": %u, \"dropped\": %u, \"status\": %d, \"total_simulated_time_ns\": %lld, "
                "\"total_render_time_ns\": %lld, \"steps_performed\": %u, \"stats_velocity\": %zu}\n",
                steps, per_frame.c_str(), fluid.num_particles(), after_first, fluid.last_add_dropped(), (int)controller.status(), (long long)controller.total_simulated_time().count(),
                (long long)controller.total_render_time().count(), controller.num_simulation_steps_performed(), fluid.pressure_solver_stats_velocity().size());
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc == 3 && !std::strcmp(argv[1], "--host-only")) return host_only(argv[2]);
        if (argc == 4 && !std::strcmp(argv[1], "--frames")) return frames(std::atoi(argv[2]), argv[3]);
        if (argc < 4) { std::fprintf(stderr, "usage: %s <scene.json> <steps> <positions.bin> [tuning=value ...] | --host-only <scene.json>\n", argv[0]); return 2; }
        const int steps = std::atoi(argv[2]);
        blub::Scene scene(argv[1]);
        blub::HybridFluid& fluid = scene.fluid_mut();
        for (int which = 0; which < 2; ++which) {      // far past convergence and no convergence DECISION: two hosts of one library then agree to rounding (tests/test_native_host.py)
            auto cfg = which ? fluid.pressure_solver_config_density() : fluid.pressure_solver_config_velocity();
            cfg->error_tolerance = 0.0f; cfg->max_num_iterations = 120; cfg->error_check_frequency = 8;
        }
        fluid.dynamic_settings()->particle_rebinning_step_frequency = 2;
        for (int a = 4; a < argc; ++a) {
            const std::string kv = argv[a];
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) { std::fprintf(stderr, "not a tuning: %s\n", argv[a]); return 2; }
            fluid.set_tuning(kv.substr(0, eq), std::atoi(kv.c_str() + eq + 1));
        }
        blub::SimulationController controller;
        const uint32_t taken = controller.fast_forward_steps(controller.simulation_delta() * steps, scene);
        fluid.synchronize();
        fluid.update_statistics();
        const std::vector<float> pos = fluid.particle_positions();
        if (FILE* f = std::fopen(argv[3], "wb")) { std::fwrite(pos.data(), sizeof(float), pos.size(), f); std::fclose(f); }
        else { std::fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
        const auto& sv = fluid.pressure_solver_stats_velocity();
        const auto& sd = fluid.pressure_solver_stats_density();
        const blub::Extent3d g = fluid.grid_dimension();
        const blub_device_views views = fluid.bind_group_renderer();
        std::printf("{\"steps_taken\": %u, \"num_particles\": %u, \"grid\": [%u, %u, %u], \"status\": %d, \"total_simulated_time_ns\": %lld, \"stats_velocity\": %zu, "
                    "\"stats_density\": %zu, \"last_velocity_iterations\": %d, \"last_velocity_error\": %.9g, \"rebinning\": %u, \"max_iterations\": %d, \"views\": %d}\n",
                    taken, fluid.num_particles(), g.width, g.height, g.depth, (int)controller.status(), (long long)controller.total_simulated_time().count(), sv.size(), sd.size(),
                    sv.empty() ? -1 : sv.back().iteration_count, sv.empty() ? 0.0 : (double)sv.back().error, fluid.dynamic_settings()->particle_rebinning_step_frequency,
                    fluid.pressure_solver_config_velocity()->max_num_iterations, (int)(views.particles_position_ll != nullptr && views.marker != nullptr && views.stream != nullptr));
        return 0;
    } catch (const blub::Error& e) {
        std::fprintf(stderr, "blub error %d: %s\n", e.status, e.what());
        return 1;
    }
}
