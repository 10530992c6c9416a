"""HIP path vs CPU oracle on the BASELINE.json configurations THEMSELVES, at full size.

  * corner_dams_256 (968 688 particles @ 256^3, the headline), dam_halfhalf and double_dam (1.2 M particles @ 128x64x64):
    every stage of step 0 compared field by field on identical inputs (the oracle drives, the engine gets the oracle's state
    before each stage), then three free-running steps with the reference's defaults: iteration counts and max|r|*dt.
  * single_cell_debug (8 particles, 64x64x128): 120 steps, positions per step.
  * the dense 2.5-D `_z` PCG mapping vs the oracle at 128^3 for k in {1, 4, 8} iterations.

Tolerances (same contract as tests/test_gpu_parity.py): marker / D1 / D2+D3 / A1 / R2+D3 / R3 bit-exact; the P2G gather
|d| <= 1e-5 max(1, |ref|), the density gather 2e-6 of the gathered density (tests/util.py DENSITY_RESIDUAL_TOL); PCG with k <= 8 fixed iterations |d| <= 3e-4 max|field| at these sizes; default solver: iteration counts
within one check interval, pressure within 3 % relative L2, max|r| as stated in _compare_solve.  The oracle is the checker here, never the thing measured.
"""
import os
import time

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _pair_from_scene(name, binning="fixed"):
    """The engine seeded from the scene JSON exactly like Scene::create_fluid_from_config, and an oracle holding the same
    particle arrays (parity tests never depend on the RNG restatement)."""
    import blub_amd
    from oracle.oracle import Oracle
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", name + ".json"))
    f = scene.fluid()
    nx, ny, nz = f.grid_dimension()
    o = Oracle(nx, ny, nz, f.num_particles() + 64)
    o.set_quirks(binning=binning)
    gravity = np.float32(list(scene.config.gravity)) / np.float32(scene.config.grid_to_world_scale)
    o.set_gravity_grid(gravity)
    o.set_particles(f.get_particles()[0])
    return scene, f, o


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _stage_on_both(o, h, stage, volumes=None):
    """volumes: what the engine takes over from the oracle before the stage (None: particles + every volume; the largest configurations
    name the stage's actual inputs -- moving nine 0.5 GB volumes through ctypes before each of ten stages would take minutes)."""
    if volumes is None:
        util.copy_state(o, h)
    else:
        util.copy_state(o, h, volumes=volumes)
    o.run_stage(stage, util.DT)
    h.run_stage(stage, util.DT)


def _compare_solve(name, o, h, which, stage, fluid, ks=(4, 8), default_solve=True):
    """One PressureSolver::solve on identical inputs (the oracle's current state): k = 4 and 8 fixed iterations -> p, r, s within
    1e-4 of their scale and identical statistics; then the reference's defaults (tolerance 0.1, 32 iterations, check every 4):
    iteration counts within one check interval; if the solve converged both errors are below the tolerance, otherwise (the
    reference's usual operating point: max|r| of an UNCONVERGED CG iterate, which is not monotone and amplifies dot-product
    rounding -- the oracle against itself with f32 instead of f64 dots moves by tens of percent) the errors agree within 2x and
    the pressure fields within 3 % in relative L2."""
    pname = "pressure_velocity" if which == 0 else "pressure_density"
    state = {v: o.read_volume(v) for v in ("residual", pname, "search")}

    def restore():
        for v, a in state.items():
            o.write_volume(v, a)
            h.write_volume(v, a)
        o.reset_pressure_cleared(which, False)
        h.mark_pressure_initialised(which, False)
    for k in ks:
        restore()
        for fl in (o, h):
            fl.set_solver_config(which, error_tolerance=0.0, max_num_iterations=k, error_check_frequency=4)
        o.run_stage(stage, util.DT)
        h.run_stage(stage, util.DT)
        for vol in (pname, "residual", "search"):
            a, b = h.read_volume(vol), o.read_volume(vol)
            scale = np.abs(b[fluid]).max()
            assert scale > 0
            util.assert_close("%s after %d iterations" % (vol, k), a[fluid], b[fluid], abs_=3e-4 * scale)   # measured worst: 1.3e-4 in one cell of 150 k (double_dam)
        assert np.all(h.read_volume(pname)[~fluid] == 0)
        (eo, io), (eh, ih) = o.solver_stats(which), h.solver_stats(which)
        assert ih == io == k and abs(eh - eo) <= 1e-4 * abs(eo) + 1e-9, ((eh, ih), (eo, io))
    if not default_solve:
        return
    restore()
    for fl in (o, h):
        fl.set_solver_config(which, error_tolerance=0.1, max_num_iterations=32, error_check_frequency=4)
    o.run_stage(stage, util.DT)
    h.run_stage(stage, util.DT)
    (eo, io), (eh, ih) = o.solver_stats(which), h.solver_stats(which)
    po, ph = o.read_volume(pname).astype(np.float64), h.read_volume(pname).astype(np.float64)
    rel_l2 = np.linalg.norm(ph - po) / max(np.linalg.norm(po), 1e-30)
    print("%s step 0 %s: oracle %d iterations / %.4g, engine %d / %.4g, pressure rel. L2 %.3g" % (name, stage, io, eo, ih, eh, rel_l2))
    assert abs(ih - io) <= 4, ((eh, ih), (eo, io))
    if io < 32 and ih < 32:
        assert eo < 0.1 and eh < 0.1
    else:
        # The oracle against ITSELF, dam_halfhalf step 0 / step 1, only the particle order inside the arrays changed (i.e. the rounding
        # of the gathers, 1e-7): velocity solve 0.3794 / 0.3794 / 0.3794 and 0.5539 / 0.5537 / 0.5524, density solve 0.953 / 0.616 /
        # 0.446 and 0.132 / 0.154 / 0.363.  The max-norm of the unconverged DENSITY residual is carried by single cells and is
        # erratic at a factor ~3; the pressure field itself is not (rel. L2 below).
        lo, hi = (0.5, 2.0) if which == 0 else (0.25, 4.0)
        assert lo < eh / eo < hi, ((eh, ih), (eo, io))
    assert rel_l2 < 3e-2, rel_l2


def _faces_reached_by_long_lists(pos, dim, cap=12):
    """Per component: the faces that take part in a P2G list longer than `cap` (transfer_gather_velocity.comp:61): WHICH particles such a list keeps is
    the order of the list atomics -- a race in the reference, ascending particle index in the oracle, lane order inside a wave and arrival order between
    waves in the engine."""
    nx, ny, nz = dim
    out = []
    for off in ((1.0, 0.5, 0.5), (0.5, 1.0, 0.5), (0.5, 0.5, 1.0)):
        d = np.floor(pos[:, :3] - np.float32(off)).astype(np.int64)
        ok = np.all((d >= 0) & (d < np.array(dim)), axis=1)
        cnt = np.bincount(((d[ok, 2] * ny + d[ok, 1]) * nx + d[ok, 0]), minlength=nx * ny * nz).reshape(nz, ny, nx)
        long_ = cnt > cap
        reach = np.zeros_like(long_)
        for kz in (0, 1):
            for ky in (0, 1):
                for kx in (0, 1):      # list d feeds the faces d + {0,1}^3
                    reach[kz:, ky:, kx:] |= long_[:nz - kz, :ny - ky, :nx - kx]
        out.append((reach, int(long_.sum()), int(cnt.max())))
    return out


def _every_stage_of_a_step(name, h, o, mid_run=False):
    """One step, stage by stage, on identical inputs: before every stage the engine takes over the oracle's state (particles with list pointers + every
    volume), both run the stage, whole volumes are compared.  mid_run: the state comes out of a running simulation, where some P2G lists exceed the
    12-entry cap -- the faces such a list reaches are compared separately (see _faces_reached_by_long_lists)."""
    t_start = time.time()
    # ---- T1-T4
    reach = _faces_reached_by_long_lists(o.get_particles()[0], h.grid_dimension()) if mid_run else None
    _stage_on_both(o, h, "transfer")
    marker = o.read_volume("marker")
    assert np.array_equal(h.read_volume("marker"), marker)
    fluid = marker == 1
    assert fluid.sum() > 100000
    for c, v in enumerate(("vel_x", "vel_y", "vel_z")):
        a, b = h.read_volume(v), o.read_volume(v)
        if not mid_run:
            util.assert_close(v, a, b, rel=1e-5)
            continue
        capped, n_long, longest = reach[c]
        # Faces of lists within the cap: the same products, added in another order (engine: per list, oracle: per round).  At step 0 the particles rest; here
        # they move at tens of cells per second, so the order of a face's up to 96 additions is worth ~1e-5 in a bad case (measured over 10 runs: 0 or 1 of
        # the ~10^6 written faces at 1.01e-5 .. 1.05e-5): util.assert_close_but_few.
        util.assert_close_but_few(v + " (faces of lists within the cap)", np.where(capped, 0, a), np.where(capped, 0, b), rel=1e-5)
        d = np.abs(a.astype(np.float64) - b)
        tol = 1e-5 * np.maximum(1.0, np.abs(b))
        bad = (d > tol) & capped
        print("%s mid-run %s: %d lists beyond the 12-entry cap (longest %d) reach %d faces; %d of those differ from the oracle's choice of 12 (%.2f %%)" % (
            name, v, n_long, longest, capped.sum(), bad.sum(), 100.0 * bad.sum() / max(1, capped.sum())))
        assert n_long > 1000      # (the state really exercises the cap; which 12 a longer list keeps is not comparable: 65 % of those faces differ, measured)
    # ---- D1 (bit-exact on identical inputs)
    _stage_on_both(o, h, "divergence")
    assert _bits_equal(h.read_volume("residual")[fluid], o.read_volume("residual")[fluid])
    assert np.abs(o.read_volume("residual")[fluid]).max() > 0
    # ---- solve #1: fixed small iteration counts (fields), then the reference's defaults (statistics + pressure field)
    _compare_solve(name, o, h, 0, "solve_velocity", fluid)
    # ---- D2 + D3
    _stage_on_both(o, h, "project")
    for v in ("vel_x", "vel_y", "vel_z"):
        assert _bits_equal(h.read_volume(v), o.read_volume(v)), v
    # ---- A1: positions, the three APIC rows, the new marker; the density list has the same cells occupied
    _stage_on_both(o, h, "advect")
    po, ph = o.get_particles(), h.get_particles()
    assert _bits_equal(ph[0][:, :3], po[0][:, :3])
    for c in (1, 2, 3):
        assert _bits_equal(ph[c], po[c]), c
    assert np.array_equal(h.read_volume("marker"), o.read_volume("marker"))
    assert np.array_equal(h.read_volume("linked_list") != 0, o.read_volume("linked_list") != 0)
    assert np.abs(po[2][:, 3]).max() > 0
    # ---- R1 (the engine walks the ORACLE's density lists here: the same 32 of a longer list on both sides)
    _stage_on_both(o, h, "density_gather")
    fluid2 = o.read_volume("marker") == 1
    util.assert_close("density residual", h.read_volume("residual")[fluid2], o.read_volume("residual")[fluid2], abs_=util.DENSITY_RESIDUAL_TOL)
    # ---- solve #2: the same three comparisons
    util.copy_state(o, h)
    _compare_solve(name, o, h, 1, "solve_density", fluid2)
    # ---- R2 + D3, R3
    _stage_on_both(o, h, "position_change")
    for v in ("vel_x", "vel_y", "vel_z"):
        assert _bits_equal(h.read_volume(v), o.read_volume(v)), v
    _stage_on_both(o, h, "correct")
    assert _bits_equal(h.get_particles()[0][:, :3], o.get_particles()[0][:, :3])
    print("%s: all stages compared in %.1f s" % (name, time.time() - t_start))


@pytest.mark.parametrize("name,particles", [("corner_dams_256", 968688), ("dam_halfhalf", 1218672), ("double_dam", 1199328)])
def test_every_stage_of_step_zero_matches_the_oracle_at_full_size(name, particles):
    scene, h, o = _pair_from_scene(name)
    try:
        assert h.num_particles() == o.num_particles == particles
        _every_stage_of_a_step(name, h, o)
    finally:
        h.close()


def test_every_stage_of_a_mid_run_step_matches_the_oracle_at_full_size():
    """Round-5 review, missing 6: every full-size identical-input comparison started at step 0 (static blocks, eight jittered particles per cell, no list
    near its cap).  Here the metric's scene runs 61 steps on the engine first -- the dams have broken and spread, bricks have gone stale and been reset,
    the particles have been re-sorted internally seven times and rebinned twice, thousands of lists exceed the 12-entry cap --, the oracle takes over the
    engine's state (particles in the caller's order, pressure fields, velocity volumes, marker) and step 61 runs stage by stage on both with the step-0
    tolerances: marker, D1, D2 + D3, A1 incl. the three APIC rows, R2 + D3, R3 bit-exact, gathers and PCG as stated there."""
    scene, h, o = _pair_from_scene("corner_dams_256")
    try:
        for _ in range(61):
            scene.step(util.DT)
        h.synchronize()
        assert h.step_counter == 61 and h.brick_counts()["fluid"] > 900      # the dams have spread (648 FLUID bricks at t = 0)
        pos, vx, vy, vz = h.get_particles()
        assert np.abs(vx[:, 3]).max() > 1.0
        o.set_particles(pos, vx, vy, vz)
        for v in ("marker", "vel_x", "vel_y", "vel_z", "pressure_velocity", "pressure_density"):
            o.write_volume(v, h.read_volume(v))
        o.reset_pressure_cleared(0, True); o.reset_pressure_cleared(1, True)
        o.step_counter = h.step_counter
        _every_stage_of_a_step("corner_dams_256 @ step 61", h, o, mid_run=True)
    finally:
        h.close()


@pytest.mark.parametrize("name,particles,dim", [("dam_halfhalf_highres", 10113264, (256, 128, 128)), ("corner_dams_512", 8065008, (512, 512, 512))])
def test_every_stage_of_step_zero_matches_the_oracle_at_the_largest_configurations(name, particles, dim):
    """BASELINE configs[3] / configs[4] (round-2 review, missing item 2): 10.1 M particles in 256x128x128 (the densest particle arrays the
    engine sees: every list at its cap) and 8.07 M particles at 512^3 (134 M cells: the sizes where 32-bit offset arithmetic and the brick
    index magic numbers are exercised hardest).  Same statements as at 256^3 -- marker, D1, D2+D3, A1 incl. the three APIC rows, R2+D3, R3
    bit-exact; gathers and fixed-iteration PCG within the stated tolerances -- with the engine taking over only each stage's inputs from
    the oracle.  At 512^3 the solves are compared after 4 fixed iterations only (an oracle iteration takes ~0.7 s there)."""
    t_start = time.time()
    scene, h, o = _pair_from_scene(name)
    big = dim[0] * dim[1] * dim[2] > 1 << 26
    try:
        assert tuple(h.grid_dimension()) == dim and h.num_particles() == o.num_particles == particles
        _stage_on_both(o, h, "transfer", volumes=())
        marker = o.read_volume("marker")
        assert np.array_equal(h.read_volume("marker"), marker)
        fluid = marker == 1
        assert fluid.sum() > 100000
        for v in ("vel_x", "vel_y", "vel_z"):
            util.assert_close(v, h.read_volume(v), o.read_volume(v), rel=1e-5)
        _stage_on_both(o, h, "divergence", volumes=("vel_x", "vel_y", "vel_z"))
        assert _bits_equal(h.read_volume("residual")[fluid], o.read_volume("residual")[fluid])
        assert np.abs(o.read_volume("residual")[fluid]).max() > 0
        _compare_solve(name, o, h, 0, "solve_velocity", fluid, ks=(4,) if big else (4, 8), default_solve=not big)
        h.set_pcg_work_mapping("rows")        # the dense 2.5-D mapping on the same system (its offsets are the ones that approach 2^32 at 512^3)
        _compare_solve(name + " (dense rows)", o, h, 0, "solve_velocity", fluid, ks=(4,), default_solve=False)
        h.set_pcg_work_mapping("auto")
        _stage_on_both(o, h, "project", volumes=("pressure_velocity",))
        for v in ("vel_x", "vel_y", "vel_z"):
            assert _bits_equal(h.read_volume(v), o.read_volume(v)), v
        _stage_on_both(o, h, "advect", volumes=())
        po, ph = o.get_particles(), h.get_particles()
        assert _bits_equal(ph[0][:, :3], po[0][:, :3])
        for c in (1, 2, 3):
            assert _bits_equal(ph[c], po[c]), c
        assert np.array_equal(h.read_volume("marker"), o.read_volume("marker"))
        assert np.array_equal(h.read_volume("linked_list") != 0, o.read_volume("linked_list") != 0)
        del po, ph
        _stage_on_both(o, h, "density_gather", volumes=("linked_list",))
        fluid2 = o.read_volume("marker") == 1
        util.assert_close("density residual", h.read_volume("residual")[fluid2], o.read_volume("residual")[fluid2], abs_=util.DENSITY_RESIDUAL_TOL)
        util.copy_state(o, h, volumes=("residual",))
        _compare_solve(name, o, h, 1, "solve_density", fluid2, ks=(4,) if big else (4, 8), default_solve=not big)
        _stage_on_both(o, h, "position_change", volumes=("pressure_density",))
        for v in ("vel_x", "vel_y", "vel_z"):
            assert _bits_equal(h.read_volume(v), o.read_volume(v)), v
        _stage_on_both(o, h, "correct", volumes=())
        assert _bits_equal(h.get_particles()[0][:, :3], o.get_particles()[0][:, :3])
        print("%s: all stages compared in %.1f s" % (name, time.time() - t_start))
    finally:
        h.close()


@pytest.mark.parametrize("name", ["corner_dams_256", "dam_halfhalf"])
def test_three_free_running_steps_stay_inside_the_oracles_own_rounding_envelope(name):
    """Reference defaults (tolerance 0.1, 32 iterations, check every 4, rebinning at step 0, Q13) on both sides, nothing
    copied between them after seeding.  What can be asserted is bounded by how the ORACLE ITSELF reacts to the rounding of
    its dot products (tests/test_oracle_kat.py::test_unconverged_cg_is_sensitive_to_dot_product_rounding; dam_halfhalf, f64 vs
    f32 accumulation, nothing else changed): step 0 identical to 4 digits, then max|r|*dt 0.554 vs 1.361 (velocity, step 1) and
    0.386 vs 1.387 (density, step 2), pressure fields 0.5 % / 4.5 % apart in relative L2, particles median 3e-4 / p99 1.6e-3 /
    max 0.24 cells.  The reference's own f32 tree reductions are a third rounding of the same kind.  Hence: step 0's velocity
    solve within 1 %; afterwards the engine's max|r| must lie within 2.5x of the interval spanned by TWO oracles stepped beside it -- f64 and
    f32 dot products, nothing else changed -- i.e. the envelope is measured at run time instead of being a fixed factor.  The DENSITY solve's
    max|r| is carried by single cells and is bimodal from run to run on the very same binary (the order of the list atomics decides: step 1 of
    corner_dams_256 measured 0.072, 0.078, 0.122, 0.232, 0.264 in five runs of round 6, 0.337 once in round 3, oracles 0.075 / 0.084), so for
    that solve the sharp statement is the FIELD: its pressure within 2 % relative L2 of the oracle's in steps 0 and 1 (measured 0.2-0.6 % in all of those
    runs; the bound was 15 %), in step 2 within max(10 %, 4x the distance of the two oracles' own fields) (3-6 %, once 12.8 %; the oracles: 4-5 %), and max|r| only within 5x of the oracles' interval;
    iteration counts may only differ while both sides hover at the tolerance, velocity pressure within 5 % (density 15 %) relative L2,
    centre of mass and occupancy histogram close."""
    from oracle.oracle import Oracle
    scene, h, o = _pair_from_scene(name)
    nx_, ny_, nz_ = h.grid_dimension()
    o32 = Oracle(nx_, ny_, nz_, h.num_particles() + 64)          # the same oracle with its dot products accumulated in f32
    o32.set_dot_mode(1)
    o32.set_gravity_grid(np.float32(list(scene.config.gravity)) / np.float32(scene.config.grid_to_world_scale))
    o32.set_particles(h.get_particles()[0])
    same_schedule = True
    try:
        for step in range(3):
            scene.step(util.DT)
            o.step(util.DT)
            o32.step(util.DT)
            h.synchronize()
            for w, hist in ((0, h.pressure_solver_stats_velocity()), (1, h.pressure_solver_stats_density())):
                eo, io = o.solver_stats(w)
                eo32 = o32.solver_stats(w)[0]
                s = hist[-1]
                pname = "pressure_velocity" if w == 0 else "pressure_density"
                po, ph = o.read_volume(pname).astype(np.float64), h.read_volume(pname).astype(np.float64)
                rel_l2 = np.linalg.norm(ph - po) / max(np.linalg.norm(po), 1e-30)
                rel_l2_oracles = np.linalg.norm(o32.read_volume(pname).astype(np.float64) - po) / max(np.linalg.norm(po), 1e-30)      # the two oracles' own distance
                print("%s step %d solver %d: oracle %d / %.4g (f32 dots: %.4g), engine %d / %.4g, pressure rel. L2 %.3g (the two oracles: %.3g)" % (
                    name, step, w, io, eo, eo32, s.iteration_count, s.error, rel_l2, rel_l2_oracles))
                assert len(hist) == step + 1
                if step == 0 and w == 0:
                    assert s.iteration_count == io and abs(s.error - eo) <= 0.01 * eo, (s, io, eo)
                env = 2.5 if w == 0 else 5.0
                assert min(eo, eo32) / env < s.error < max(eo, eo32) * env, (step, w, s, io, eo, eo32)
                if s.iteration_count != io:
                    assert max(s.error, eo) < 0.4, (step, w, s, io, eo)       # both hover around the tolerance of 0.1
                # A solve that stops at an earlier check than the other side's leaves a visibly different iterate (7 % measured) and
                # from then on the two particle systems are two different trajectories (the next density solve differed by 69 %):
                # fields are only compared while every solve so far ran the same number of iterations on both sides.
                if s.iteration_count != io:
                    same_schedule = False
                if same_schedule:
                    # (density field: 0.2-0.6 % in steps 0 / 1; step 2 is the third capped solve of a free run: 3-6 % in most runs, 12.8 % once in 25 -- the
                    #  two ORACLES are 4-5 % apart there, so the bound follows their distance, measured in this very run, instead of a constant)
                    floor = 0.05 if w == 0 else (0.02 if step < 2 else 0.10)
                    assert rel_l2 < max(floor, 4.0 * rel_l2_oracles), (step, w, rel_l2, rel_l2_oracles)
        # permutation-invariant particle metrics after three steps (binning orders differ inside a cell)
        a, b = h.get_particles()[0][:, :3].astype(np.float64), o.get_particles()[0][:, :3].astype(np.float64)
        assert a.shape == b.shape
        assert np.abs(a.mean(0) - b.mean(0)).max() < 2e-3
        nx, ny, nz = h.grid_dimension()
        occ = lambda p: np.bincount(((p[:, 2].astype(int) * ny + p[:, 1].astype(int)) * nx + p[:, 0].astype(int)), minlength=nx * ny * nz)
        l1 = np.abs(occ(a) - occ(b)).sum() / len(a)
        print("%s after 3 steps: occupancy L1 %.4g, centre of mass %.3g" % (name, l1, np.abs(a.mean(0) - b.mean(0)).max()))
        assert l1 < 0.05
    finally:
        h.close()


def test_single_cell_debug_tracks_the_oracle_for_120_steps():
    """BASELINE configs[0]: 8 particles in cell (31, 31, 63) fall, hit the floor and spread.
    (a) per step from the SAME state (the engine restarts every step from the oracle's particles and pressure fields):
        positions within 1e-5 cells, every one of the 120 steps;
    (b) free running, nothing copied: within 1e-5 cells while the particles fall freely (measured: bit-identical); the deviation
        after the impact is reported only (8 particles bouncing off a wall are a chaotic system)."""
    scene, h, o = _pair_from_scene("single_cell_debug", binning="off")
    import blub_amd
    h2 = blub_amd.HybridFluid(h.grid_dimension(), 64, binning="off")
    try:
        assert h.num_particles() == 8
        h.particle_rebinning_step_frequency = 0
        h2.set_gravity_grid(np.float32(list(scene.config.gravity)) / np.float32(scene.config.grid_to_world_scale))
        worst_sync, worst_free, worst_free_fall = 0.0, 0.0, 0.0
        y0 = o.get_particles()[0][:, 1].copy()
        for step in range(120):
            # (a) h2 := oracle state, one step on both
            pos, vx, vy, vz = o.get_particles()
            h2.set_particles(pos, vx, vy, vz)
            for v in ("pressure_velocity", "pressure_density"):
                h2.write_volume(v, o.read_volume(v))
            h2.mark_pressure_initialised(0, True); h2.mark_pressure_initialised(1, True)
            h2.step_counter = o.step_counter
            o.step(util.DT)
            h2.step(util.DT)
            h.step(util.DT)
            po = o.get_particles()[0][:, :3]
            d_sync = np.abs(h2.get_particles()[0][:, :3] - po).max()
            d_free = np.abs(h.get_particles()[0][:, :3] - po).max()
            worst_sync = max(worst_sync, float(d_sync))
            worst_free = max(worst_free, float(d_free))
            falling = po[:, 1].min() > 2.5
            if falling:
                worst_free_fall = max(worst_free_fall, float(d_free))
            assert d_sync <= 1e-5, "step %d: %g cells from the same state" % (step, d_sync)
        print("single_cell_debug: worst per-step deviation %.3g, free running %.3g while falling / %.3g overall" % (worst_sync, worst_free_fall, worst_free))
        assert o.get_particles()[0][:, 1].max() < y0.min() - 20       # they did fall to the floor
        assert worst_free_fall <= 1e-5
        # after the impact the 8-particle system is chaotic (measured: per-step deviations of <= 8e-6 grow to tens of cells within
        # 90 steps on both sides alike); the free-running engine only has to stay inside the domain
        pf = h.get_particles()[0][:, :3]
        assert np.all(np.isfinite(pf)) and np.all(pf >= 1.001) and np.all(pf <= np.float32(h.grid_dimension()) - np.float32(1.001))
    finally:
        h.close()
        h2.close()


@pytest.mark.parametrize("iters", [1, 4, 8])
def test_dense_z_mapping_matches_the_oracle_at_128_cubed(iters):
    """The 2.5-D dense mapping (k_pcg_*_z, the roofline path) against the oracle itself -- not only self-consistency: SOLID shell,
    FLUID interior with an AIR layer under the lid and a SOLID block inside, b = sin sin sin, warm start p != 0."""
    import blub_amd
    from oracle.oracle import Oracle
    n = 128
    h = blub_amd.HybridFluid((n, n, n), 16, binning="off")
    try:
        h.set_pcg_work_mapping("rows")
        o = Oracle(n, n, n, 16)
        marker = np.zeros((n, n, n), np.int8)
        marker[1:-1, 1:-1, 1:-1] = 1
        marker[1:-1, -2, 1:-1] = -1
        marker[40:50, 20:30, 60:90] = 0
        fluid = marker == 1
        ax = np.sin(2 * np.pi * (np.arange(n) + 0.5) / n).astype(np.float32)
        b = (ax[:, None, None] * ax[None, :, None] * ax[None, None, :]).astype(np.float32)
        b[~fluid] = 0
        rng = np.random.default_rng(3)
        p0 = np.where(fluid, rng.standard_normal((n, n, n)) * 0.05, 0).astype(np.float32)
        for fl in (o, h):
            fl.write_volume("marker", marker)
            fl.write_volume("residual", b)
            fl.write_volume("pressure_velocity", p0)
            fl.set_solver_config(0, error_tolerance=0.0, max_num_iterations=iters, error_check_frequency=4)
        o.reset_pressure_cleared(0, True)      # warm start: neither side clears the pressure field (pressure_solver.rs:601-603)
        h.mark_pressure_initialised(0, True)
        o.run_stage("solve_velocity", util.DT)
        h.run_stage("solve_velocity", util.DT)
        for vol in ("pressure_velocity", "residual", "search"):
            a, c = h.read_volume(vol), o.read_volume(vol)
            scale = np.abs(c[fluid]).max()
            util.assert_close("%s after %d iterations" % (vol, iters), a[fluid], c[fluid], abs_=1e-4 * scale)
        assert np.all(h.read_volume("pressure_velocity")[~fluid] == 0)
        (eo, io), (eh, ih) = o.solver_stats(0), h.solver_stats(0)
        assert ih == io == iters and abs(eh - eo) <= 1e-4 * abs(eo), ((eh, ih), (eo, io))
    finally:
        h.close()


def _divergence_d1(marker, vx, vy, vz):
    """divergence_compute.comp:59-84 on FLUID cells, vectorised, without moving solids (none of the BASELINE scenes has any): the flux of the
    face velocities, where a face shared with a SOLID cell contributes nothing (v_wall - (v_wall - v_solid) with v_solid = 0).  Summation in
    the shader's order, in f32."""
    m = np.pad(marker, 1, constant_values=0)
    pad = lambda a: np.pad(a, 1)
    X, Y, Z = pad(vx), pad(vy), pad(vz)
    c = (slice(1, -1),) * 3
    sh = lambda a, dz, dy, dx: a[1 + dz:a.shape[0] - 1 + dz, 1 + dy:a.shape[1] - 1 + dy, 1 + dx:a.shape[2] - 1 + dx]
    px, py, pz = X[c], Y[c], Z[c]
    qx, qy, qz = sh(X, 0, 0, -1), sh(Y, 0, -1, 0), sh(Z, -1, 0, 0)
    div = (px - qx).astype(np.float32)
    div = div + (py - qy)
    div = div + (pz - qz)
    zero = np.float32(0)
    div = div + np.where(sh(m, 0, 0, -1) == 0, qx, zero)
    div = div + np.where(sh(m, 0, -1, 0) == 0, qy, zero)
    div = div + np.where(sh(m, -1, 0, 0) == 0, qz, zero)
    div = div - np.where(sh(m, 0, 0, 1) == 0, px, zero)
    div = div - np.where(sh(m, 0, 1, 0) == 0, py, zero)
    div = div - np.where(sh(m, 1, 0, 0) == 0, pz, zero)
    return np.where(marker == 1, div, zero).astype(np.float32)


@pytest.mark.parametrize("name", ["corner_dams_256", "dam_halfhalf", "double_dam"])
def test_divergence_left_after_the_pressure_projection_is_the_reported_residual(name):
    """north_star's parity quantity as SURVEY 8c(ii) defines it, on the GPU: D1's formula (divergence_compute.comp:59-84) applied to the
    ENGINE's velocity volumes right after D2 (divergence_remove.comp), max|.| over FLUID cells -- held against (a) the engine's own
    statistic max|r| * dt of the solve that produced the pressure (in exact arithmetic they are the same number: r = b - A p), (b) the
    engine's residual volume cell by cell, (c) the same quantity of the oracle stepping the same particles."""
    from oracle.oracle import Oracle
    scene, h, o = _pair_from_scene(name, binning="off")
    # the same oracle with its dot products accumulated in f32: the free-running step below is held to the interval the TWO oracles span (the envelope test's rule)
    nx_, ny_, nz_ = h.grid_dimension()
    o32 = Oracle(nx_, ny_, nz_, h.num_particles() + 64)
    o32.set_quirks(binning="off")
    o32.set_dot_mode(1)
    o32.set_gravity_grid(np.float32(list(scene.config.gravity)) / np.float32(scene.config.grid_to_world_scale))
    o32.set_particles(h.get_particles()[0])
    try:
        h.particle_rebinning_step_frequency = 0
        for step in range(2):
            for st in ("transfer", "divergence", "solve_velocity", "project"):
                h.run_stage(st, util.DT)
                o.run_stage(st, util.DT)
                o32.run_stage(st, util.DT)
            e_h, it_h = h.solver_stats(0)
            e_o, it_o = o.solver_stats(0)
            marker = h.read_volume("marker")
            fluid = marker == 1
            div = _divergence_d1(marker, h.read_volume("vel_x"), h.read_volume("vel_y"), h.read_volume("vel_z"))
            r = np.where(fluid, h.read_volume("residual"), 0).astype(np.float32)
            scale = max(1.0, float(np.abs(h.read_volume("vel_y")).max()))
            # (b) the recurrence's residual IS the divergence that is left, cell by cell (rounding of ~32 updates of values of size `scale`)
            assert np.abs(div - r).max() <= 2e-4 * scale, (np.abs(div - r).max(), scale)
            # (a) and its max-norm is what the solver reports
            dmax = float(np.abs(div).max()) * util.DT
            assert abs(dmax - e_h) <= 2e-4 * scale * util.DT + 1e-3 * e_h, (dmax, e_h)
            # (c) the oracle's velocity field after ITS projection, same formula
            div_o = _divergence_d1(o.read_volume("marker"), o.read_volume("vel_x"), o.read_volume("vel_y"), o.read_volume("vel_z"))
            dmax_o = float(np.abs(div_o).max()) * util.DT
            print("%s step %d: max|div| dt after D2: engine %.4g (reports %.4g after %d iterations), oracle %.4g (reports %.4g after %d)" % (name, step, dmax, e_h, it_h, dmax_o, e_o, it_o))
            assert abs(dmax_o - e_o) <= 2e-4 * scale * util.DT + 1e-3 * e_o
            if step == 0:
                assert it_h == it_o and abs(dmax - dmax_o) <= 0.02 * dmax_o, ((dmax, it_h), (dmax_o, it_o))
            else:   # free-running: an unconverged CG amplifies the rounding of its dots -- dam_halfhalf: oracle 0.554 with f64 dots, 1.36 with f32 dots, engine 0.55 .. 1.41 from run to run
                div_o32 = _divergence_d1(o32.read_volume("marker"), o32.read_volume("vel_x"), o32.read_volume("vel_y"), o32.read_volume("vel_z"))
                dmax_o32 = float(np.abs(div_o32).max()) * util.DT
                print("%s step %d: the oracle with f32 dot products: %.4g" % (name, step, dmax_o32))
                assert abs(it_h - it_o) <= 4 and min(dmax_o, dmax_o32) / 2.5 < dmax < max(dmax_o, dmax_o32) * 2.5, (dmax, dmax_o, dmax_o32)
            for st in ("advect", "density_gather", "solve_density", "position_change", "correct"):
                h.run_stage(st, util.DT)
                o.run_stage(st, util.DT)
                o32.run_stage(st, util.DT)
            h.step_counter = step + 1
            o.step_counter = step + 1
    finally:
        h.close()
