"""A host of the engine in a compiled language (include/blub_hybrid_fluid.hpp: the reference's `HybridFluid` / `Scene` / `SimulationController` surface in
C++ above the C-ABI, standing in for the Rust shim this image cannot compile -- SURVEY.md 8f-1): tests/native/hybrid_fluid_host.cpp is the reference's main
loop written against it.  CPU: it builds with g++, links against the in-tree library, parses scenes and reports the NO_DEVICE status as an exception.
GPU: it fast-forwards a scene and ends where the Python mirror of the same surface ends."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import util
from tests.conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "hybrid_fluid_host.cpp")
SCENE = os.path.join(ROOT, "scenes", "corner_dams_128.json")


def _build(tmp_path):
    import blub_amd
    lib_dir = os.path.dirname(blub_amd.lib_path())
    exe = str(tmp_path / "hybrid_fluid_host")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + lib_dir, "-lblubhip",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], timeout=300)
    return exe


def test_the_cpp_host_builds_links_and_reports_errors_as_exceptions(tmp_path):
    """No GPU needed: the header compiles warning-free as C++17, every symbol it uses resolves against libblubhip.so, the host-only entry points work
    (scene JSON, the cube generator) and the engine's statuses arrive as blub::Error -- NO_DEVICE from the constructor on a box without a GPU (there is no
    CPU path to fall back to), IO for a missing scene file."""
    import blub_amd
    exe = _build(tmp_path)
    out = json.loads(subprocess.run([exe, "--host-only", SCENE], capture_output=True, text=True, timeout=120, check=True).stdout)
    cfg = blub_amd.Scene.parse(path=SCENE).config
    assert out["grid"] == list(cfg.grid_dimension) and out["cubes"] == cfg.num_fluid_cubes
    scale = cfg.grid_to_world_scale
    lo = [(cfg.cube_min[0][k] - cfg.world_position[k]) / scale for k in range(3)]
    hi = [(cfg.cube_max[0][k] - cfg.world_position[k]) / scale for k in range(3)]
    assert out["seeded"] == len(blub_amd.seed_fluid_cube(tuple(cfg.grid_dimension), cfg.max_num_particles, 0, lo, hi)) > 10000
    assert out["create_status"] == (0 if has_gpu() else -7)
    assert out["missing_scene_status"] == -5
    assert out["version"].startswith("blubhip")


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("scene_name", ["corner_dams_128", "wavegenerator_cube"])      # (the second: a static object, animated and voxelised before every step by Scene::step)
def test_the_cpp_host_and_the_python_mirror_end_in_the_same_state(tmp_path, scene_name):
    """Scene::new -> SimulationController::fast_forward_steps(4 steps) -> accessors, once from C++ and once from Python: the same number of steps, the
    same clocks and statistics history, the same particles (matched one to one by position; solves of a fixed 120 iterations, far past convergence and without a convergence DECISION --
    tools/host_pace_probe.py, profiles/r06_host_pace_probe_*.json: two runs of one library agree to the rounding of
    the gathers' list order), the settings written through the `&mut` proxies arrive."""
    import blub_amd
    from blub_amd.simulation_controller import SimulationController
    steps = 4      # (corner_dams_128: in step 5 a rounding-level event -- NOT a convergence decision, not the host's pace: tools/host_pace_probe.py -- splits the runs of ONE host into two families 3e-3 apart at p99.9)
    SCENE = os.path.join(ROOT, "scenes", scene_name + ".json")
    exe = _build(tmp_path)
    binfile = str(tmp_path / "positions.bin")
    # (which of its bit-different P2G gathers the engine takes is decided from brick counts as they land on the host -- a speed choice, and corner_dams_128
    #  sits at its threshold: both hosts pin it)
    res = subprocess.run([exe, SCENE, str(steps), binfile, "p2g_compact=1"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout)
    sc = blub_amd.Scene(path=SCENE)
    try:
        f = sc.fluid()
        for w in (0, 1):
            f.set_solver_config(w, error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)
        f.particle_rebinning_step_frequency = 2
        f.set_tuning("p2g_compact", 1)
        ctl = SimulationController()
        taken = ctl.fast_forward_steps(sc, ctl.simulation_delta_ns * steps)
        f.synchronize()
        f.update_statistics()
        pos = f.get_particles()[0]
        assert out["steps_taken"] == taken == steps
        assert out["num_particles"] == f.num_particles() == len(pos)
        assert out["grid"] == list(f.grid_dimension())
        assert SimulationController._STATUS[out["status"]] == ctl.status == "Paused" and out["total_simulated_time_ns"] == ctl.total_simulated_time_ns
        assert out["stats_velocity"] == len(f.pressure_solver_stats_velocity()) and out["stats_density"] == len(f.pressure_solver_stats_density())
        assert out["stats_velocity"] >= steps - 3 and out["stats_velocity"] > 0          # (samples still in flight at the last poll are allowed to be missing)
        if scene_name == "wavegenerator_cube":      # (the moving cube has entered the domain: solid cells inside the domain shell)
            assert (f.read_volume("marker")[1:-1, 1:-1, 1:-1] == 0).sum() > 1000
        assert out["rebinning"] == 2 and out["max_iterations"] == 120 and out["views"] == 1
        native = np.fromfile(binfile, np.float32).reshape(-1, 4)
        assert native.shape == pos.shape
        # (the rebinning of steps 0 and 2 re-defines the caller's order, and the order INSIDE a cell is the order of its atomics -- a race between any two
        #  runs: particles are matched by position, one to one, not by index)
        from scipy.spatial import cKDTree
        d, idx = cKDTree(pos[:, :3].astype(np.float64)).query(native[:, :3].astype(np.float64), k=1)
        assert len(np.unique(idx)) == len(pos), "matching is not one-to-one"
        fell = sc.reset().get_particles()[0][:, 1].mean() - pos[:, 1].mean()
        print("C++ host vs Python mirror after %d steps: median %.3g p99.9 %.3g max %.3g cells (the centre of mass fell %.3g cells)" % (steps, np.median(d), np.quantile(d, 0.999), d.max(), fell))
        assert fell > 0.01
        assert np.median(d) < 2e-5 and np.quantile(d, 0.999) < 5e-4 and util.max_but_three(d) < 2e-2 and d.max() < 1.0
    finally:
        sc.fluid().close()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_a_fluid_built_by_hand_and_driven_frame_by_frame(tmp_path):
    """HybridFluid::new + add_fluid_cube x 2 (the second truncated at max_num_particles) + set_gravity_grid, then main.rs's loop without a renderer: two frames of a
    fixed 1/60 s, on_frame_submitted + frame_steps.  The C++ host and the Python mirror take the same steps in the same frames (two per frame at 120 steps per
    second), end with the same clocks, counts and particles (solves of a fixed 120 iterations; the particles matched one to one by position)."""
    import blub_amd
    from blub_amd.simulation_controller import SimulationController
    from scipy.spatial import cKDTree
    frames = 2
    exe = _build(tmp_path)
    binfile = str(tmp_path / "positions.bin")
    res = subprocess.run([exe, "--frames", str(frames), binfile], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout)
    f = blub_amd.HybridFluid((64, 48, 32), 120000)
    try:
        f.add_fluid_cube((2.0, 2.0, 2.0), (30.0, 34.0, 30.0))
        after_first = f.num_particles()
        f.add_fluid_cube((40.0, 2.0, 4.0), (62.0, 20.0, 28.0))
        f.set_gravity_grid((0.0, -981.0, 0.0))
        for w in (0, 1):
            f.set_solver_config(w, error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)
        ctl = SimulationController()
        per_frame = []
        for _ in range(frames):
            ctl.on_frame_submitted(16666667)
            per_frame.append(ctl.frame_steps_fluid(f))
        f.synchronize()
        f.update_statistics()
        assert out["steps_per_frame"] == per_frame and out["steps_taken"] == sum(per_frame) == ctl.num_simulation_steps_performed == out["steps_performed"] >= 3
        assert out["after_first_cube"] == after_first and out["num_particles"] == f.num_particles() == 120000 and out["dropped"] == f.last_add_dropped() > 0
        assert SimulationController._STATUS[out["status"]] == ctl.status
        assert out["total_simulated_time_ns"] == ctl.total_simulated_time_ns and out["total_render_time_ns"] == ctl.total_render_time_ns
        assert abs(ctl.total_render_time_ns - frames * 16666667) <= 4 * frames      # (Timer::on_frame_submitted scales by time_scale through f32 seconds: Duration::mul_f32, timer.rs:80)
        pos = f.get_particles()[0]
        native = np.fromfile(binfile, np.float32).reshape(-1, 4)
        assert native.shape == pos.shape
        d, idx = cKDTree(pos[:, :3].astype(np.float64)).query(native[:, :3].astype(np.float64), k=1)
        assert len(np.unique(idx)) == len(pos), "matching is not one-to-one"
        print("C++ host vs Python mirror, %d frames = %d steps: median %.3g p99.9 %.3g max %.3g cells" % (frames, sum(per_frame), np.median(d), np.quantile(d, 0.999), d.max()))
        # (measured over 8 runs: median 0 -- most particles bit-equal --, p99.9 8e-5 .. 2.6e-4, max 4e-4 .. 1.3e-3: the blocks fall from rest, 120 000 particles hit the cap's edge)
        assert np.median(d) < 2e-5 and np.quantile(d, 0.999) < 2e-3 and util.max_but_three(d) < 5e-2 and d.max() < 1.0
    finally:
        f.close()
