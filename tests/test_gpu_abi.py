"""Error behaviour and bookkeeping of the C-ABI on a real device (the reference's unwrap()/assert!/error! paths)."""
import os

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def test_create_rejects_unsupported_grids():
    import blub_amd
    from blub_amd.hybrid_fluid import BlubError
    for dim, status in (((30, 32, 32), -2),      # x not a multiple of 4 (float4 rows)
                        ((16, 16, 16), -2),      # <= 16384 cells: pressure_solver.rs:551 asserts the same
                        ((2, 64, 64), -1)):
        with pytest.raises(BlubError) as e:
            blub_amd.HybridFluid(dim, 16)
        assert e.value.status == status, dim


def test_add_fluid_cube_truncates_and_counts_like_the_reference():
    import blub_amd
    f = blub_amd.HybridFluid((32, 32, 32), 1000)
    try:
        f.add_fluid_cube((1, 1, 1), (5, 5, 5))            # 4*4*4*8 = 512
        assert f.num_particles() == 512 and f.last_add_dropped() == 0
        f.add_fluid_cube((8, 8, 8), (16, 16, 16))         # would be 4096: truncated to the 488 left (hybrid_fluid.rs:627-633)
        assert f.num_particles() == 1000 and f.last_add_dropped() == 4096 - 488     # what the reference's error! reports
        p = f.get_particles()[0]
        assert np.all(p[:512, :3] >= 1) and np.all(p[:512, :3] <= 5) and np.all(p[512:, :3] >= 8)
        assert np.all(p.view(np.uint32)[:, 3] == 0xFFFFFFFF)
    finally:
        f.close()


def test_step_argument_checks_and_statistics_ring():
    import blub_amd
    from blub_amd.hybrid_fluid import BlubError
    f = blub_amd.HybridFluid((32, 32, 32), 4096)
    try:
        with pytest.raises(BlubError) as e:
            f.step(0.0)
        assert e.value.status == -1
        with pytest.raises(BlubError):
            f.set_particles(np.zeros((5000, 3), np.float32))
        f.add_fluid_cube((4, 4, 4), (12, 8, 12))
        f.set_gravity_grid((0, -981.0, 0))
        assert f.pressure_solver_config_velocity().max_num_iterations == 32 and abs(f.pressure_solver_config_density().error_tolerance - 0.1) < 1e-7
        assert f.particle_rebinning_step_frequency == 60
        for _ in range(130):                                # more than the 100-sample history (pressure_solver.rs:101)
            f.step(util.DT)
            f.update_statistics()
        f.synchronize()
        sv, sd = f.pressure_solver_stats_velocity(), f.pressure_solver_stats_density()
        assert len(sv) == 100 and len(sd) == 100
        assert all(0 < s.iteration_count <= 32 for s in sv)
        assert f.step_counter == 130
        views = f.bind_group_renderer()
        assert all(views[k] for k in ("particles_position_ll", "velocity_x", "marker", "pressure_from_density", "stream"))
    finally:
        f.close()


def test_scene_json_to_fluid_matches_manual_construction():
    import os
    import blub_amd
    from tests.conftest import ROOT
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "single_cell_debug.json"))
    f = scene.fluid()
    try:
        assert f.grid_dimension() == (64, 64, 128) and f.num_particles() == 8
        p0 = f.get_particles()[0][:, :3].copy()
        g = float(np.float32(-9.81) / np.float32(0.01))
        for n in range(1, 4):                               # free fall: y_n = y_0 + g dt^2 n(n+1)/2 (tests/test_oracle_kat.py)
            scene.step(util.DT)
        f.synchronize()
        p = f.get_particles()[0][:, :3]
        # step 0 rebins (Q13): compare as sets, sorted by x (x and z do not change)
        a, b = p[np.argsort(p[:, 0])], p0[np.argsort(p0[:, 0])]
        assert np.abs(a[:, 1] - (b[:, 1] + g * util.DT * util.DT * 6)).max() < 2e-4
        assert np.abs(a[:, [0, 2]] - b[:, [0, 2]]).max() < 1e-5
    finally:
        f.close()


def test_profile_trace_and_chrome_trace(tmp_path):
    """Observability parity: per-launch timeline with the reference's scope labels as chrome-trace JSON."""
    import json
    import os
    import blub_amd
    from blub_amd.hybrid_fluid import write_chrome_trace
    from tests.conftest import ROOT
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_128.json"))
    f = scene.fluid()
    try:
        scene.step(util.DT)
        f.profile_enable(True)
        f.profile_reset()
        for _ in range(2):
            scene.step(util.DT)
        ev = f.profile_trace()
        f.profile_enable(False)
        names = {e["name"] for e in ev}
        assert {"build_lists", "gather_velocity", "advect", "correct", "extrapolate"} <= names
        assert "pcg_iter" in names or {"pcg_dir", "pcg_update"} <= names      # single-reduction (default) or reference schedule
        assert len(ev) > 100 and all(e["duration_us"] > 0 for e in ev)
        starts = [e["start_us"] for e in ev]
        assert starts == sorted(starts) and starts[0] == 0.0
        assert {e["step"] for e in ev} == {1, 2}
        per = f.profile_read()
        kname = "pcg_iter" if "pcg_iter" in per else "pcg_dir"
        assert abs(sum(e["duration_us"] for e in ev if e["name"] == kname) - per[kname]["total_ms"] * 1e3) < 1.0
        path = tmp_path / "simulation-trace.json"
        n = write_chrome_trace(f, str(path))
        doc = json.load(open(path))
        assert n == len(ev) and len(doc["traceEvents"]) > n
        assert any(e["name"] == "primary pressure solver (divergence)" for e in doc["traceEvents"])
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):
            write_chrome_trace(f, os.path.join(out, "simulation-trace_corner_dams_128_2steps.json"))
    finally:
        f.close()


def test_checkpoint_and_resume_through_the_state_exchange_calls():
    """SURVEY 5 "Checkpoint / resume" (absent in the reference, asked of the new engine): particles + APIC rows, the two pressure
    fields (the solver's warm start, pressure_init.comp:50-83) and the step counter are the complete state.  A run resumed from
    them in a NEW handle continues like the uninterrupted one (up to the engine's own run-to-run noise; converged solves)."""
    import blub_amd
    dim = (48, 32, 32)
    pos, vel, maxp = util.make_dam(*dim, fill=(0.5, 0.6, 1.0), seed=3)
    cfg = dict(error_tolerance=2e-6, max_num_iterations=400, error_check_frequency=8)

    def fresh():
        f = blub_amd.HybridFluid(dim, maxp)
        f.set_gravity_grid((0.0, -981.0, 0.0))
        for w in (0, 1):
            f.set_solver_config(w, **cfg)
        f.particle_rebinning_step_frequency = 0   # (binning permutes particles inside a cell in atomic order: keep indices comparable)
        return f
    a = fresh()
    b = None
    try:
        a.set_particles(pos, *vel)
        for _ in range(4):
            a.step(util.DT)
        state = {"particles": a.get_particles(), "step_counter": a.step_counter,
                 "pv": a.read_volume("pressure_velocity"), "pd": a.read_volume("pressure_density")}
        b = fresh()
        b.set_particles(state["particles"][0][:, :3], *state["particles"][1:])
        b.write_volume("pressure_velocity", state["pv"])
        b.write_volume("pressure_density", state["pd"])
        b.mark_pressure_initialised(0, True)
        b.mark_pressure_initialised(1, True)
        b.step_counter = state["step_counter"]
        for _ in range(4):
            a.step(util.DT)
            b.step(util.DT)
        assert a.step_counter == b.step_counter == 8
        pa, pb = a.get_particles(), b.get_particles()
        d = np.abs(pa[0][:, :3] - pb[0][:, :3]).max(axis=1)
        print("resume vs uninterrupted after 4 more steps: median %.3g p99 %.3g max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()))
        assert np.median(d) < 1e-4 and np.quantile(d, 0.99) < 3e-3 and d.max() < 0.1
        assert np.abs(pa[1] - pb[1]).max() < 5.0
    finally:
        a.close()
        if b is not None:
            b.close()


def test_native_scheduler_fast_forwards_the_fluid_through_the_c_abi():
    """`SimulationController::fast_forward_steps` (simulation_controller.rs:96-157) as blub_controller_* -- batches of 16
    blub_fluid_step + a wait, at least one step, wall clock kept; no Python in the stepping loop."""
    import blub_amd
    from blub_amd.simulation_controller import NS, SimulationController
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_128.json"))
    f = scene.fluid()
    c = SimulationController()
    try:
        n = c.fast_forward_steps_fluid(f, 40 * c.simulation_delta_ns)
        assert n == 40 and f.step_counter == 40 and c.total_simulated_time_ns == 40 * c.simulation_delta_ns
        assert c.status == SimulationController.PAUSED and c.computation_time_last_fast_forward > 0
        assert len(f.pressure_solver_stats_velocity()) == 40            # update_statistics after every step + the waits: every sample landed
        print("fast-forward of 40 steps of corner_dams_128: %.1f steps/s" % (n / c.computation_time_last_fast_forward))
        assert c.fast_forward_steps_fluid(f, 1) == 1 and f.step_counter == 41    # "jump at least one simulation step" (:119-121)
        # the same controller drives a whole Scene (Python callbacks): one 60 Hz frame = two steps
        c.pause_or_resume()
        c.on_frame_submitted(NS // 60 + 1000)
        assert c.frame_steps(scene) == 2 and f.step_counter == 43
        pos = f.get_particles()[0]
        assert np.all(np.isfinite(pos[:, :3]))
    finally:
        c.close()
        f.close()
