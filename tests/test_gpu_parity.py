"""Parity of the HIP path (through the C-ABI) with the CPU oracle, stage by stage and for whole steps.

Tolerances (written here):
  * element-wise kernels on identical inputs (divergence, project, advect, position_change, correct): BIT-EXACT
    (same f32 operation order, -ffp-contract=off on both sides)
  * gathers: transfer |d| <= 1e-5 * max(1, |ref|); density_gather 2e-6 of the gathered density = 2e-4 on the residual it
    writes (tests/util.py DENSITY_RESIDUAL_TOL)   -- only the summation order differs
  * PCG, fixed k <= 8 iterations: p, r, s |d| <= 1e-4 * max|field|; default config: residual norm within 5 %, pressure
    within 3 % relative L2 (unconverged CG iterates amplify dot-product rounding -- see test_pcg_default_config)
  * whole step, binning off, converged solves: particle positions |d| <= 1e-4 cells; default solver: median < 1e-4,
    p99 < 5e-3, max < 0.15 cells (32 fixed iterations)
"""
import numpy as np
import pytest

from tests import util
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

GRID = (48, 40, 32)


@pytest.fixture()
def pair():
    pos, vel, maxp = util.make_dam(*GRID)
    o, h = util.new_pair(*GRID, maxp)
    o.set_particles(pos, *vel)
    h.set_particles(pos, *vel)
    yield o, h
    h.close()


def run_until(o, stage):
    for s in util.STEP_ORDER:
        if s == stage:
            return
        if s != "binning":
            o.run_stage(s, util.DT)


def test_transfer(pair):
    o, h = pair
    o.run_stage("transfer", util.DT)
    h.run_stage("transfer", util.DT)
    assert np.array_equal(h.read_volume("marker"), o.read_volume("marker"))
    for v in ("vel_x", "vel_y", "vel_z"):
        util.assert_close(v, h.read_volume(v), o.read_volume(v), rel=1e-5)
    assert np.abs(o.read_volume("vel_y")).max() > 1.0


@pytest.mark.parametrize("stage,outputs", [("divergence", ["residual"]), ("project", ["vel_x", "vel_y", "vel_z"]),
                                           ("position_change", ["vel_x", "vel_y", "vel_z"])])
def test_elementwise_grid_stage_bit_exact(pair, stage, outputs):
    o, h = pair
    run_until(o, stage)
    util.copy_state(o, h)
    o.run_stage(stage, util.DT)
    h.run_stage(stage, util.DT)
    for v in outputs:
        a, b = h.read_volume(v), o.read_volume(v)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s differs in %d cells" % (v, (a != b).sum())
    assert any(np.abs(o.read_volume(v)).max() > 0 for v in outputs)


@pytest.mark.parametrize("mapping", ["rows", "bricks", "bricks_single"])
@pytest.mark.parametrize("iters", [0, 1, 4, 7, 8])
def test_pcg_fixed_iterations(pair, iters, mapping):
    """Fixed iteration count (tolerance 0): p, r, s after k iterations. Only the dot-product summation order differs;
    before the rounding noise is amplified by many unconverged CG iterations the fields agree to 1e-4 of their scale."""
    o, h = pair
    util.set_mapping(h, mapping)
    for f in (o, h):
        f.set_solver_config(0, error_tolerance=0.0, max_num_iterations=iters, error_check_frequency=4)
    run_until(o, "solve_velocity")
    util.copy_state(o, h)
    o.run_stage("solve_velocity", util.DT)
    h.run_stage("solve_velocity", util.DT)
    fluid = o.read_volume("marker") == 1
    for name in ("pressure_velocity", "residual", "search"):
        a, b = h.read_volume(name), o.read_volume(name)
        scale = np.abs(b[fluid]).max()
        assert scale > 0
        util.assert_close(name, a[fluid], b[fluid], abs_=1e-4 * scale)
    assert np.all(h.read_volume("pressure_velocity")[~fluid] == 0)   # pressure_init.comp:45-48
    eo, io = o.solver_stats(0)
    eh, ih = h.solver_stats(0)
    assert ih == io == iters
    assert abs(eh - eo) <= 1e-4 * abs(eo) + 1e-9


@pytest.mark.parametrize("mapping", ["rows", "bricks", "bricks_single"])
@pytest.mark.parametrize("which,stage", [(0, "solve_velocity"), (1, "solve_density")])
def test_pcg_default_config(pair, which, stage, mapping):
    """The reference's operating point (32 iterations, check every 4) stops far from convergence (max|r| ~ 12), where
    the CG iterate is sensitive to the rounding of the dot products (the oracle itself moves by 2 % when its dots are
    accumulated in f32 instead of f64).  So: pressure within 3 % in relative L2, residual max-norm within 2x, and the
    reported state must be self-consistent: r == b - A p recomputed in f64.  (Tolerance 0 pins the iteration count: with
    0.1 this scene sits at 0.0995 after 28 iterations, a coin flip between 28 and 32.)"""
    o, h = pair
    util.set_mapping(h, mapping)
    for f in (o, h):
        f.set_solver_config(which, error_tolerance=0.0, max_num_iterations=32, error_check_frequency=4)
    run_until(o, stage)
    util.copy_state(o, h)
    b = o.read_volume("residual").astype(np.float64)
    o.run_stage(stage, util.DT)
    h.run_stage(stage, util.DT)
    name = "pressure_velocity" if which == 0 else "pressure_density"
    marker = o.read_volume("marker")
    fluid = marker == 1
    po, ph = o.read_volume(name).astype(np.float64), h.read_volume(name).astype(np.float64)
    assert np.all(ph[~fluid] == 0)
    rel_l2 = np.linalg.norm(ph - po) / np.linalg.norm(po)
    assert rel_l2 < 3e-2, rel_l2
    eo, io = o.solver_stats(which)
    eh, ih = h.solver_stats(which)
    # (max|r| of the unconverged DENSITY solve is carried by single cells: the oracle against itself spreads by ~3x when only the
    #  rounding of its inputs changes, tests/test_gpu_baseline_parity.py::_compare_solve; the pressure field above is the robust measure)
    # (round-2 ADVICE: the reference-order mappings keep the tight bound; only the single-reduction opt-in gets the wide one for the density solve)
    lo, hi = (0.25, 4.0) if (which == 1 and mapping == "bricks_single") else (0.5, 2.0)
    print("%s solver %d: max|r| dt engine %.4g oracle %.4g, pressure rel. L2 %.3g" % (mapping, which, eh, eo, rel_l2))
    assert ih == io == 32 and lo < eh / eo < hi, ((eh, ih), (eo, io))
    # self-consistency of the HIP state: r = b - A p (pressure.glsl:34-75), A from the marker
    mpad = np.pad(marker, 1, constant_values=0)
    ppad = np.pad(ph * fluid, 1)
    diag = np.zeros_like(ph)
    nb = np.zeros_like(ph)
    for ax in range(3):
        for sft in (-1, 1):
            diag += np.roll(mpad, sft, ax)[1:-1, 1:-1, 1:-1] != 0
            nb += np.roll(ppad, sft, ax)[1:-1, 1:-1, 1:-1] * (np.roll(mpad, sft, ax)[1:-1, 1:-1, 1:-1] == 1)
    r_expected = (b - (diag * ph - nb)) * fluid
    r_hip = h.read_volume("residual").astype(np.float64) * fluid
    assert np.abs(r_hip - r_expected).max() <= 2e-4 * max(1.0, np.abs(b).max())
    assert abs(np.abs(r_hip).max() * util.DT - eh) <= 1e-5 * eh + 1e-9   # reported error = max|r| * dt (pressure_solver.rs:162)


@pytest.mark.parametrize("grid", [8, 100000])
def test_brick_list_kernels_do_not_depend_on_their_launch_grid(pair, grid):
    """The kernels that loop over a brick list are launched with an ESTIMATED grid (the list lengths live on the device): far fewer workgroups
    than bricks (every workgroup loops) and far more (clamped to the brick count; the surplus exits) must give the same bits."""
    o, h = pair
    h.set_tuning("list_launch_grid", grid)
    o.run_stage("transfer", util.DT)
    h.run_stage("transfer", util.DT)
    for v in ("vel_x", "vel_y", "vel_z"):
        util.assert_close(v, h.read_volume(v), o.read_volume(v), rel=1e-5)
    for stage, outputs in (("divergence", ["residual"]), ("project", ["vel_x", "vel_y", "vel_z"]), ("position_change", ["vel_x", "vel_y", "vel_z"])):
        run_until(o, stage)
        util.copy_state(o, h)
        o.run_stage(stage, util.DT)
        h.run_stage(stage, util.DT)
        for v in outputs:
            a, b = h.read_volume(v), o.read_volume(v)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s / %s differs in %d cells" % (stage, v, (a != b).sum())


@pytest.mark.parametrize("schedule", ["reference", "single_reduction"])
def test_divergence_formed_inside_the_solve_is_the_same_solve(pair, schedule):
    """Inside blub_fluid_step the brick-mapped velocity solve forms b = div u in its init kernel (k_pcg_init_b<true>) instead of reading the residual
    volume a divergence kernel wrote.  Same function, same operands: residual, pressure, search direction and the solver statistics must agree
    BIT FOR BIT with the two-kernel sequence ("fuse_divergence" = 2 defers the divergence across blub_fluid_run_stage calls as well; the
    marker-derived lists of the stage hook are deterministic)."""
    o, h = pair
    run_until(o, "divergence")
    fluid = o.read_volume("marker") == 1
    h.set_pcg_work_mapping("bricks")
    h.set_pcg_schedule(schedule)
    out = []
    for fuse in (0, 2):
        h.set_tuning("fuse_divergence", fuse)
        util.copy_state(o, h)
        res = o.read_volume("residual").copy()
        res[fluid] = np.nan                          # whatever the residual volume holds in FLUID cells is not an input of either sequence
        h.write_volume("residual", res)
        h.run_stage("divergence", util.DT)
        h.run_stage("solve_velocity", util.DT)
        out.append(([h.read_volume(v) for v in ("residual", "pressure_velocity", "search")], h.solver_stats(0)))
    for a, b, name in zip(out[0][0], out[1][0], ("residual", "pressure", "search")):
        assert np.array_equal(a[fluid].view(np.uint32), b[fluid].view(np.uint32)), "%s differs in %d FLUID cells" % (name, (a[fluid] != b[fluid]).sum())
    assert out[0][1] == out[1][1] and out[0][1][1] > 0
    assert np.abs(out[0][0][1]).max() > 0


def test_advect_bit_exact(pair):
    o, h = pair
    run_until(o, "advect")
    util.copy_state(o, h)
    o.run_stage("advect", util.DT)
    h.run_stage("advect", util.DT)
    po, ph = o.get_particles(), h.get_particles()
    assert np.array_equal(ph[0][:, :3].view(np.uint32), po[0][:, :3].view(np.uint32))
    for c in (1, 2, 3):
        assert np.array_equal(ph[c].view(np.uint32), po[c].view(np.uint32))
    assert np.array_equal(h.read_volume("marker"), o.read_volume("marker"))
    n = o.num_particles
    assert util.lists_as_sets(h.read_volume("linked_list"), ph[0], n) == util.lists_as_sets(o.read_volume("linked_list"), po[0], n)
    assert np.abs(po[1][:, 3]).max() > 0


def test_density_gather(pair):
    o, h = pair
    run_until(o, "density_gather")
    util.copy_state(o, h)
    o.run_stage("density_gather", util.DT)
    h.run_stage("density_gather", util.DT)
    util.assert_close("residual", h.read_volume("residual"), o.read_volume("residual"), abs_=util.DENSITY_RESIDUAL_TOL)
    assert np.abs(o.read_volume("residual")).max() > 0


def test_correct_bit_exact(pair):
    o, h = pair
    run_until(o, "correct")
    util.copy_state(o, h)
    before = o.get_particles()[0].copy()
    o.run_stage("correct", util.DT)
    h.run_stage("correct", util.DT)
    po, ph = o.get_particles()[0], h.get_particles()[0]
    assert np.array_equal(ph[:, :3].view(np.uint32), po[:, :3].view(np.uint32))
    assert np.abs(po[:, :3] - before[:, :3]).max() > 0


def test_binning_is_cell_ordered_permutation(pair):
    o, h = pair
    h2 = None
    try:
        import blub_amd
        pos, vel, maxp = util.make_dam(*GRID, seed=5)
        rng = np.random.default_rng(0)
        perm = rng.permutation(pos.shape[0])
        h2 = blub_amd.HybridFluid(GRID, maxp, binning="fixed")
        h2.set_particles(pos[perm])
        h2.run_stage("binning", util.DT)
        got = h2.get_particles()[0][:, :3]
        # permutation: multiset of 12-byte records is unchanged
        a = np.sort(got.view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=("x", "y", "z"))
        b = np.sort(pos.view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=("x", "y", "z"))
        assert np.array_equal(a, b)
        # cell-contiguous in linear (x fastest) order: particle_binning_prefixsum.comp:17-22
        cell = got.astype(np.int64)
        lin = (cell[:, 2] * GRID[1] + cell[:, 1]) * GRID[0] + cell[:, 0]
        assert np.all(np.diff(lin) >= 0)
    finally:
        if h2 is not None:
            h2.close()


def test_literal_binning_mode_matches_the_oracles_literal_mode():
    """BLUB_BINNING_LITERAL: Q4 as the reference's three shaders run it (particle_binning_count.comp:9-13 without an i < NumParticles guard,
    1-based destinations in _rewrite_particles.comp:8-16, whole-buffer copy hybrid_fluid.rs:885-891), against the oracle's `literal` mode
    (itself checked against a numpy emulation of the shaders, tests/test_oracle_crosscheck.py).  The slot a particle gets inside its cell is
    an atomic race in the reference and in the engine (ascending index in the oracle), so records are compared as multisets per cell; the
    cell at which the live range ends may keep different members of its particles on the two sides -- only its COUNT is compared."""
    import blub_amd
    from oracle.oracle import Oracle
    pos, vel, maxp = util.make_dam(*GRID, seed=7)
    rng = np.random.default_rng(3)
    pos = pos[rng.permutation(pos.shape[0])][:pos.shape[0] - 37]          # not a multiple of 64
    n = pos.shape[0]
    pad = (n + 63) // 64 * 64 - n
    o = Oracle(*GRID, n + 200)
    o.set_quirks(binning="literal")
    h = blub_amd.HybridFluid(GRID, n + 200, binning="literal")
    try:
        for rounds in range(2):            # the second pass bins what the first one left behind the live range as well
            if rounds == 0:
                o.set_particles(pos)
                h.set_particles(pos)
            o.run_stage("binning", util.DT)
            h.run_stage("binning", util.DT)
            po, ph = o.get_particles()[0][:, :3], h.get_particles()[0][:, :3]
            assert po.shape == ph.shape == (n, 3)
            lin = lambda a: ((a[:, 2].astype(np.int64) * GRID[1] + a[:, 1].astype(np.int64)) * GRID[0] + a[:, 0].astype(np.int64))
            co, ch = lin(po), lin(ph)
            # slot 0 is never written (it keeps what it held), slots 1 .. n - 1 are cell-contiguous in linear order on both sides
            assert np.array_equal(ph[0], po[0])
            assert np.all(np.diff(ch[1:]) >= 0) and np.all(np.diff(co[1:]) >= 0)
            uo, cnt_o = np.unique(co, return_counts=True)
            uh, cnt_h = np.unique(ch, return_counts=True)
            assert np.array_equal(uo, uh) and np.array_equal(cnt_o, cnt_h)          # every cell keeps the same NUMBER of records
            last_cell = co[-1]                                                        # the cell the live range ends in
            rec = lambda a, c: np.sort(np.ascontiguousarray(a[c != last_cell]).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=("x", "y", "z"))
            assert np.array_equal(rec(ph, ch), rec(po, co))
            if rounds == 0:
                # pad zero records were binned into the live range and pad + 1 real particles fell off its end
                assert int((np.abs(ph).sum(axis=1) == 0).sum()) == pad + 1               # slot 0 (never written) + the pad zero records
                orig = np.sort(np.ascontiguousarray(pos[:, :3]).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=("x", "y", "z"))
                kept = np.sort(np.ascontiguousarray(ph).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=("x", "y", "z"))
                assert len(orig) - np.isin(orig, kept).sum() == pad + 1
    finally:
        h.close()


@pytest.mark.parametrize("mapping", ["auto", "rows"])
def test_full_step_converged_solver(pair, mapping):
    """With both pressure solves run to convergence the solution no longer depends on CG rounding: one whole step
    (binning off, identical particle order) reproduces the oracle's particle positions to 1e-4 cells."""
    o, h = pair
    util.set_mapping(h, mapping)
    cfg = dict(error_tolerance=2e-6, max_num_iterations=400, error_check_frequency=8)
    for f in (o, h):
        f.set_solver_config(0, **cfg)
        f.set_solver_config(1, **cfg)
    o.step(util.DT)
    h.step(util.DT)
    po, ph = o.get_particles(), h.get_particles()
    d = np.abs(ph[0][:, :3] - po[0][:, :3]).max(axis=1)
    assert (d > 1e-4).mean() < 1e-4, "fraction of particles off by > 1e-4 cells: %g (max %g)" % ((d > 1e-4).mean(), d.max())
    for c in (1, 2, 3):
        util.assert_close("particle velocity rows", ph[c], po[c], abs_=2e-2)
    assert h.solver_stats(0)[1] < 400 and o.solver_stats(0)[1] < 400


def test_full_step_loose_solver(pair):
    """The reference's operating point stops the CG far from convergence (32 iterations, max|r| ~ 10), where the iterate
    amplifies rounding: on the CPU alone, re-ordering the linked lists or accumulating the dots in f32 moves particles by
    median 1e-4 / p99 4e-4 / max 0.06 cells after one step.  Parity is therefore a distribution.  (Tolerance 0 pins the
    iteration count at 32: with the default tolerance this scene sits at 0.0995 vs 0.1 at iteration 28, a coin flip.)"""
    o, h = pair
    for f in (o, h):
        for w in (0, 1):
            f.set_solver_config(w, error_tolerance=0.0, max_num_iterations=32, error_check_frequency=4)
    o.step(util.DT)
    h.step(util.DT)
    po, ph = o.get_particles(), h.get_particles()
    d = np.abs(ph[0][:, :3] - po[0][:, :3]).max(axis=1)
    print("deviation quantiles (cells): median %.3g  p99 %.3g  max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()))
    # Twelve runs of this test over rounds 5 and 6 (same binaries run to run: the order of the list atomics seeds the rounding that 32 unconverged iterations
    # amplify): median 3.3e-5, 8.4e-5, 8.5e-5, 1.05e-4 (x3) ... 8.65e-4, 1.21e-3; p99 1.9e-4 ... 4.5e-3; max 0.03 ... 0.11.  The tail of that spread sat ON the
    # former bound (1e-3 / 5e-3: one run in ten failed); the bound is 2.5x the worst run -- a defect shows at 1e-2 and above, and the tight statement for the same step
    # is test_full_step_converged_solver (1e-4 cells).
    # (the maximum is ONE particle at the surface or a wall that takes another branch: 0.03 .. 0.155 over ~75 runs; all but the three worst particles are held to the bound,
    #  those three to one cell)
    assert np.median(d) < 3e-3 and np.quantile(d, 0.99) < 1.2e-2 and np.partition(d, len(d) - 4)[len(d) - 4] < 0.15 and d.max() < 1.0
    for w in (0, 1):
        eo, io = o.solver_stats(w)
        eh, ih = h.solver_stats(w)
        assert ih == io == 32
        # max|r| of an unconverged CG is not monotone (0.197, 0.175, 0.242, 0.140 at i = 8..20 here); the density solve follows the engine's own advection of this step
        # and is carried by single cells: the full-size tests' factor (tests/test_gpu_baseline_parity.py)
        assert (0.5 < eh / eo < 2.0) if w == 0 else (0.25 < eh / eo < 4.0), (w, eh, eo)


def test_convergence_decision_semantics(pair):
    """pressure_reduce.comp:82-94 / pressure_solver.rs:672-697: the first checked iteration (multiples of the check
    frequency) whose max|r| is below tolerance/dt is reported, later work is disabled."""
    o, h = pair
    run_until(o, "solve_velocity")
    util.copy_state(o, h)
    state = {v: o.read_volume(v) for v in ("residual", "pressure_velocity", "search")}
    # tolerance placed between the errors of two consecutive checks of the oracle, so rounding cannot flip the decision
    errs = {}
    for it in range(4, 68, 4):
        for v, a in state.items():
            o.write_volume(v, a)
        o.reset_pressure_cleared(0, False)
        o.set_solver_config(0, error_tolerance=0.0, max_num_iterations=it, error_check_frequency=4)
        o.run_stage("solve_velocity", util.DT)
        errs[it] = o.solver_stats(0)[0]
    # first check iteration whose error drops clearly (x0.7) below everything seen before: the tolerance goes in the gap
    hi = next(it for it in range(8, 68, 4) if errs[it] < 0.7 * min(errs[j] for j in range(4, it, 4)))
    tol = float(np.sqrt(errs[hi] * min(errs[j] for j in range(4, hi, 4))))
    for f in (o, h):
        f.set_solver_config(0, error_tolerance=tol, max_num_iterations=64, error_check_frequency=4)
    for v, a in state.items():
        o.write_volume(v, a)
        h.write_volume(v, a)
    o.reset_pressure_cleared(0, False)
    h.mark_pressure_initialised(0, False)
    o.run_stage("solve_velocity", util.DT)
    h.run_stage("solve_velocity", util.DT)
    eo, io = o.solver_stats(0)
    eh, ih = h.solver_stats(0)
    assert io == hi and ih == hi and eh < tol and abs(eh - eo) < 0.2 * eo
    fluid = o.read_volume("marker") == 1
    a, b = h.read_volume("pressure_velocity"), o.read_volume("pressure_velocity")
    util.assert_close("pressure", a[fluid], b[fluid], abs_=2e-3 * np.abs(b).max())


def test_multi_step_statistics():
    """5 steps of a dam break (default solver), with and without rebinning on the HIP side: permutation-invariant metrics."""
    import blub_amd
    pos, vel, maxp = util.make_dam(*GRID, velocity_scale=0.0)
    o, h = util.new_pair(*GRID, maxp, binning="off")
    h2 = blub_amd.HybridFluid(GRID, maxp, binning="fixed")
    try:
        h2.set_gravity_grid((0.0, -981.0, 0.0))
        h2.particle_rebinning_step_frequency = 2
        for f in (o, h, h2):
            f.set_particles(pos)
        for _ in range(5):
            for f in (o, h, h2):
                f.step(util.DT)
        ref = o.get_particles()[0][:, :3].astype(np.float64)
        for name, f in (("binning off", h), ("binning on", h2)):
            got = f.get_particles()[0][:, :3].astype(np.float64)
            assert got.shape == ref.shape
            assert np.all(got >= 1.001 - 1e-6) and np.all(got <= np.array(GRID) - 1.001 + 1e-6)
            com = np.abs(got.mean(0) - ref.mean(0)).max()
            occ = lambda p: np.bincount(((p[:, 2].astype(int) * GRID[1] + p[:, 1].astype(int)) * GRID[0] + p[:, 0].astype(int)), minlength=np.prod(GRID))
            l1 = np.abs(occ(got) - occ(ref)).sum() / ref.shape[0]
            print("%s: centre-of-mass diff %.3g cells, occupancy L1 %.3g" % (name, com, l1))
            # 5 steps of a chaotic particle system: two CPU oracles that differ only in dot-product rounding are already
            # 2.1e-3 cells / 0.023 apart in these metrics
            assert com < 2e-2 and l1 < 0.1, "%s: com %g, occupancy histogram L1 distance %g" % (name, com, l1)
    finally:
        h.close()
        h2.close()


def test_lod0_preconditioner_mode():
    pos, vel, maxp = util.make_dam(*GRID)
    o, h = util.new_pair(*GRID, maxp, precond="lod0", solver=dict(error_tolerance=0.0, max_num_iterations=6, error_check_frequency=4))
    try:
        o.set_particles(pos, *vel)
        run_until(o, "solve_velocity")
        util.copy_state(o, h)
        o.run_stage("solve_velocity", util.DT)
        h.run_stage("solve_velocity", util.DT)
        po, ph = o.read_volume("pressure_velocity"), h.read_volume("pressure_velocity")
        util.assert_close("pressure lod0", ph, po, abs_=1e-4 * np.abs(po).max())
        assert h.solver_stats(0)[1] == o.solver_stats(0)[1]
    finally:
        h.close()


def test_solid_voxels_and_scene_single_cell():
    """Solid voxel input (stand-in for SceneVoxelization) + the reference's single_cell_debug scene."""
    import blub_amd
    nx, ny, nz = GRID
    pos, vel, maxp = util.make_dam(*GRID, fill=(0.4, 0.5, 1.0))
    o, h = util.new_pair(*GRID, maxp)
    try:
        vox = np.zeros((nz, ny, nx, 4), np.float32)
        vox[4:12, 1:6, 26:34, 3] = 1.0          # a static block partly inside the fluid
        vox[4:12, 1:6, 26:34, 0] = 0.5          # moving in +x
        o.write_volume("solid", vox)
        h.set_solid_voxels(vox)
        o.set_particles(pos, *vel)
        h.set_particles(pos, *vel)
        for f in (o, h):
            for w in (0, 1):
                f.set_solver_config(w, error_tolerance=2e-6, max_num_iterations=600, error_check_frequency=8)
        o.step(util.DT)
        h.step(util.DT)
        po, ph = o.get_particles()[0], h.get_particles()[0]
        d = np.abs(ph[:, :3] - po[:, :3]).max(axis=1)
        print("solid scene deviation: median %.3g p99 %.3g max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()))
        assert (d > 1e-4).mean() < 1e-3 and d.max() < 0.05
        assert np.array_equal(h.read_volume("marker"), o.read_volume("marker")) or (h.read_volume("marker") != o.read_volume("marker")).mean() < 1e-4
    finally:
        h.close()


def test_sparse_bricks_track_moving_fluid():
    """A blob falling through a tall domain leaves its bricks behind: every volume the renderer can see must still equal
    the dense reference's (velocity / pressure 0 and marker AIR where the fluid used to be) -- this exercises the
    stale-brick clearing of the brick-sparse work lists (blub_bricks.hip.h)."""
    dim = (32, 64, 32)
    rng = np.random.default_rng(3)
    cells = np.stack(np.meshgrid(np.arange(10, 20), np.arange(48, 58), np.arange(10, 20), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    o, h = util.new_pair(*dim, pos.shape[0], solver=dict(error_tolerance=2e-6, max_num_iterations=300, error_check_frequency=8))
    try:
        o.set_particles(pos)
        h.set_particles(pos)
        for step in range(30):
            o.step(util.DT)
            h.step(util.DT)
            if step in (0, 14, 29):
                mh, mo = h.read_volume("marker"), o.read_volume("marker")
                bad = np.argwhere(mh != mo)
                assert len(bad) == 0, "step %d: %d marker cells differ, first (z,y,x)=%s hip=%s oracle=%s; y range %s; brick counts %s" % (
                    step, len(bad), bad[:5].tolist(), mh[tuple(bad[:5].T)].tolist(), mo[tuple(bad[:5].T)].tolist(),
                    (bad[:, 1].min(), bad[:, 1].max()), h.brick_counts())
                for v in ("vel_x", "vel_y", "vel_z", "pressure_velocity", "pressure_density"):
                    a, b = h.read_volume(v), o.read_volume(v)
                    util.assert_close("%s after step %d" % (v, step), a, b, abs_=2e-3 * max(1.0, np.abs(b).max()))
                    assert np.array_equal(a != 0, b != 0) or ((a != 0) != (b != 0)).mean() < 1e-4, (v, step)
        bc = h.brick_counts()
        assert 0 < bc["fluid"] < bc["active"] < bc["total"]
        po, ph = o.get_particles()[0][:, :3], h.get_particles()[0][:, :3]
        assert po[:, 1].mean() < 40          # it really fell out of its initial bricks (8 cells high)
        assert np.abs(ph - po).max() < 5e-3
    finally:
        h.close()


def _match_particles(a, b):
    """Nearest-neighbour matching of two particle sets (the slab exchange permutes the order)."""
    from scipy.spatial import cKDTree
    d, idx = cKDTree(b).query(a, k=1)
    assert len(np.unique(idx)) == len(a), "matching is not one-to-one"
    return d


@pytest.mark.parametrize("slabs,async_exchange,transport", [(2, True, "host"), (3, True, "host"), (3, False, "host"), (2, True, "direct"), (3, True, "direct"), (8, True, "direct")])
def test_z_slab_decomposition_matches_single_domain(slabs, async_exchange, transport):
    """the z-slab protocol (ghost particles, halo planes, all-reduced PCG scalars, migration) run as `slabs`
    slabs on ONE GPU (loopback transport) reproduces the single-domain engine.  The blob straddles the slab interfaces
    and shears across them, so every exchange carries data.
    The engine is not bit-reproducible run to run (the linked lists are built with atomic exchanges, their order changes the
    rounding of the gathers, and the particle system amplifies that: two identical single-domain runs are already
    p99.9 = 5e-3 / max 0.02 cells apart after two steps of this scene).  The test therefore measures that noise floor with a
    second single-domain instance (printed) and requires the slab group to stay inside the envelope of that noise: the
    floor is bimodal (a particle within 1e-4 of a cell boundary -- ~40 of them here -- lands in the other cell and now and
    then that flips a surface marker: step-1 maxima are either ~1e-3 or ~0.025 cells, rerun vs rerun), so the bounds are
    absolute: step 0 (every exchange except migration already feeds it) p99 < 4e-4 / max < 3e-3 cells; later steps
    median < 2e-4, p99 < 3e-3, p99.9 < 0.03, max < 0.1 cells."""
    import blub_amd
    dim = (32, 32, 48)
    rng = np.random.default_rng(4)
    cells = np.stack(np.meshgrid(np.arange(6, 26), np.arange(8, 20), np.arange(6, 42), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    vel = [np.zeros((pos.shape[0], 4), np.float32) for _ in range(3)]
    vel[2][:, 3] = 6.0 * np.sin(pos[:, 0] * 0.4)            # z-velocities push particles across the interfaces
    # fixed 120 iterations (tolerance 0): far past convergence and without a convergence DECISION, which would otherwise
    # make the run-to-run noise bimodal (stopping at check 24 vs 32 moves particles by up to 0.026 cells)
    cfg = dict(error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    rerun = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=slabs, binning="off")
    assert group.transport() == "direct"          # the default of a local group since round 4
    group.set_transport(transport)
    group.set_async_exchange(async_exchange)
    if transport == "direct":       # (solves of more than 64 iterations would otherwise run the reference's two-kernel order: its exchanges stay host-issued pushes)
        for i in range(slabs):
            group.local_fluid(i).set_tuning("pcg1_max_iterations", 1000)
        single.set_tuning("pcg1_max_iterations", 1000)
        rerun.set_tuning("pcg1_max_iterations", 1000)
    try:
        for f in (single, rerun, group):
            f.set_gravity_grid((0.0, -981.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        ranges = [group.local_range(i) for i in range(slabs)]
        assert ranges[0][0] == 0 and ranges[-1][1] == dim[2] and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        counts0 = [group.local_fluid(i).num_particles() for i in range(slabs)]
        assert sum(counts0) == pos.shape[0] and all(c > 0 for c in counts0[1:-1])      # (eight slabs: the outermost two start empty)
        ops0 = group.transport_ops()
        for step in range(3):
            for f in (single, rerun, group):
                f.step(util.DT)
            ps = single.get_particles()[0][:, :3].astype(np.float64)
            pr = rerun.get_particles()[0][:, :3].astype(np.float64)
            pg = group.get_particles()[0][:, :3].astype(np.float64)
            assert pg.shape == ps.shape                                   # no particle lost or duplicated
            floor = np.abs(pr - ps).max(axis=1)
            d = _match_particles(pg, ps)                                   # also asserts the matching is one-to-one
            q = lambda a: (np.median(a), np.quantile(a, 0.99), np.quantile(a, 0.999), a.max())
            print("step %d  z-slab(%d) vs single: median %.3g p99 %.3g p99.9 %.3g max %.3g | rerun noise floor: %.3g %.3g %.3g %.3g" % ((step, slabs) + q(d) + q(floor)))
            bounds = (3e-5, 4e-4, 1.5e-3, 3e-3) if step == 0 else (2e-4, 3e-3, 3e-2, 0.1)
            for a, b in zip(q(d), bounds):
                assert a <= b, (step, q(d), bounds)
        counts1 = [group.local_fluid(i).num_particles() for i in range(slabs)]
        assert sum(counts1) == pos.shape[0]
        assert counts1 != counts0, "no particle migrated: the test does not exercise the exchange"
        # host synchronisations by particle exchanges: four in the first step (no history to size the messages from), none afterwards --
        # or four per step with the round-2 protocol; the direct transport never synchronises the host, not for exchanges and not to look at a
        # solve's `done`, and issues 12 transport operations per step (the 120 PCG iterations of each solve exchange from inside their kernels)
        if transport == "direct":
            assert group.host_syncs() == (0, 0), group.host_syncs()
            print("direct transport: %d transport operations in 3 steps" % (group.transport_ops() - ops0))
            assert group.transport_ops() - ops0 == 3 * 12, group.transport_ops() - ops0
            assert group.held_back() == 0
        else:
            assert group.host_syncs()[0] == (4 if async_exchange else 12), group.host_syncs()
        # every slab only holds particles of its own z-range
        pgl = group.get_particles()[0]
        off = 0
        for i, (z0, z1) in enumerate(ranges):
            z = pgl[off:off + counts1[i], 2]
            off += counts1[i]
            assert np.all(z >= z0) and np.all(z < z1)
        # the marker of the union of slabs equals the single-domain one up to the cells chaotic particles flip
        m_single = single.read_volume("marker")
        m_group = np.zeros_like(m_single)
        for i, (z0, z1) in enumerate(ranges):
            m_group[z0:z1] = group.local_fluid(i).read_volume("marker")[z0:z1]
        assert (m_group != m_single).mean() < 2e-3
        for w in (0, 1):   # identical solver statistics on every slab (the scalars are all-reduced)
            st = [group.local_fluid(i).solver_stats(w) for i in range(slabs)]
            assert all(x == st[0] for x in st)
    finally:
        single.close()
        rerun.close()
        group.close()


@pytest.mark.parametrize("schedule", ["single_reduction", "reference"])
def test_z_slab_solve_follows_convergence(schedule):
    """The slab solve launches iterations through the check that ended the previous solve, looks at `done` and extends by
    one check interval at a time (blub_slab.inc.hip: slab_solve).  With the reference's solver defaults the group must report
    the same iteration counts as the single-domain engine (within one check interval: the dot products are summed in a
    different order), stay inside the run-to-run noise envelope, and issue far fewer transport operations than a
    full-length solve once it has a previous iteration count to go by."""
    import blub_amd
    dim = (32, 32, 48)
    rng = np.random.default_rng(5)
    cells = np.stack(np.meshgrid(np.arange(4, 28), np.arange(2, 14), np.arange(4, 44), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=3, binning="off")
    group.set_transport("host")       # (this test is about the host-issued operations of the copy / RCCL protocol)
    # transport operations of one solve with k launched iterations: init exchange (+ the w_0 exchange of the single-reduction
    # schedule), per iteration ONE grouped operation (single reduction) or TWO (reference schedule), the pressure halo
    solve_ops = (lambda k: 2 + k + 1) if schedule == "single_reduction" else (lambda k: 1 + 2 * k + 1)
    try:
        for f in (single, group):
            f.set_pcg_schedule(schedule)
            f.set_gravity_grid((0.0, -9.81 / 0.01, 0.0))
            f.set_particles(pos)
        ops = []
        for step in range(6):
            before = group.transport_ops()
            single.step(util.DT)
            group.step(util.DT)
            group.synchronize()
            ops.append(group.transport_ops() - before)
        single.synchronize()
        single.update_statistics()
        slab_fluids = [group.local_fluid(i) for i in range(3)]
        for f in slab_fluids:
            f.update_statistics()
        full = 1 + 4 * 2 + 5 + 2 * solve_ops(33)      # brick-count gather, particle exchanges (counts + payload), velocity halos, 2 solves of 33 iterations
        assert ops[0] == full, (ops, full)           # first step: no previous iteration count
        hist = lambda f, w: [x.iteration_count for x in (f.pressure_solver_stats_velocity() if w == 0 else f.pressure_solver_stats_density())]
        its = []
        for w in (0, 1):
            it_s = hist(single, w)
            it_all = [hist(f, w) for f in slab_fluids]
            assert all(x == it_all[0] for x in it_all)           # every slab takes the same decisions
            it_g = it_all[0]
            its.append(it_g)
            print("solve %d iterations: single %s  slabs %s  transport ops/step %s" % (w, it_s, it_g, ops))
            assert len(it_g) == 6 and len(it_s) == 6 and all(abs(a - b) <= 4 for a, b in zip(it_s, it_g)), (it_s, it_g)
            assert all(0 < x <= 32 for x in it_g)
        for step in range(1, 6):   # launched per solve = iterations through the later of {previous, this} deciding check + its detection
            # (from the second step on a particle exchange is ONE grouped operation: fixed-capacity messages with a count header)
            need = 1 + 4 * 1 + 5 + sum(solve_ops(min(33, max(its[w][step], its[w][step - 1]) + 2)) for w in (0, 1))
            assert ops[step] == need, (step, ops, need, its)
        d = _match_particles(group.get_particles()[0][:, :3].astype(np.float64), single.get_particles()[0][:, :3].astype(np.float64))
        assert np.median(d) < 1e-3 and np.quantile(d, 0.99) < 2e-2, (np.median(d), np.quantile(d, 0.99), d.max())
    finally:
        single.close()
        group.close()


def test_one_launch_and_two_kernel_list_builds_agree():
    """k_bricks_build (classification + scatter in one launch, the blocks waiting for each other's counts) against the two-kernel scan it
    replaces on grids with at most one brick block per CU: identical brick counts after every build of three steps and -- with the
    solves converged, so that nothing amplifies the order of the atomics -- the same particles to 1e-4 cells."""
    import blub_amd
    pos, vel, maxp = util.make_dam(*GRID, seed=3)
    out = []
    for two_kernel in (0, 1):
        h = blub_amd.HybridFluid(GRID, maxp, binning="off")
        try:
            h.set_tuning("bricks_two_kernel_build", two_kernel)
            h.set_gravity_grid((0.0, -981.0, 0.0))
            h.set_particles(pos, *vel)
            for w in (0, 1):
                h.set_solver_config(w, error_tolerance=2e-6, max_num_iterations=400, error_check_frequency=8)
            counts = []
            for _ in range(3):
                h.step(util.DT)
                counts.append(h.brick_counts())
            out.append((counts, h.get_particles()[0][:, :3].astype(np.float64), h.read_volume("marker")))
        finally:
            h.close()
    assert out[0][0] == out[1][0] and out[0][0][-1]["fluid"] > 0
    assert np.array_equal(out[0][2], out[1][2])
    d = np.abs(out[0][1] - out[1][1]).max(axis=1)
    assert (d > 1e-4).mean() < 1e-3, d.max()


def test_partial_bricks_odd_grid_full_step():
    """Grid dimensions that are not multiples of the 16x8x4 brick (only x % 4 == 0 is required): partial bricks at the
    upper domain faces, full step against the oracle with converged solves."""
    dim = (36, 30, 22)
    pos, vel, maxp = util.make_dam(*dim, fill=(0.7, 0.6, 1.0), seed=11)
    o, h = util.new_pair(*dim, maxp, solver=dict(error_tolerance=0.0, max_num_iterations=150, error_check_frequency=8))
    try:
        o.set_particles(pos, *vel)
        h.set_particles(pos, *vel)
        for _ in range(2):
            o.step(util.DT)
            h.step(util.DT)
        po, ph = o.get_particles()[0][:, :3], h.get_particles()[0][:, :3]
        d = np.abs(ph - po).max(axis=1)
        print("odd grid: median %.3g p99 %.3g max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()))
        assert np.median(d) < 2e-4 and np.quantile(d, 0.99) < 3e-3 and d.max() < 0.1
        assert (h.read_volume("marker") != o.read_volume("marker")).mean() < 2e-3
        bc = h.brick_counts()
        assert bc["total"] == 3 * 4 * 6
    finally:
        h.close()


def test_empty_fluid_steps_are_harmless():
    import blub_amd
    h = blub_amd.HybridFluid((32, 32, 32), 128)
    try:
        h.set_gravity_grid((0, -981.0, 0))
        for _ in range(3):
            h.step(util.DT)
        h.synchronize()
        assert h.num_particles() == 0
        for v in ("vel_x", "vel_y", "vel_z", "pressure_velocity", "pressure_density"):
            a = h.read_volume(v)
            assert np.all(a == 0) and np.all(np.isfinite(a))
        m = h.read_volume("marker")
        assert np.all(m[1:-1, 1:-1, 1:-1] == -1) and np.all(m[0] == 0) and np.all(m[:, 0] == 0) and np.all(m[:, :, -1] == 0)
        assert h.solver_stats(0)[1] in range(0, 33)
    finally:
        h.close()


@pytest.mark.parametrize("schedule", ["reference", "single_reduction"])
@pytest.mark.parametrize("first", [1, 3, 9])
def test_pcg_persistent_tail_kernel(first, schedule):
    """Brick-mapped solves hand the iterations the host did not launch to ONE persistent kernel (k_pcg_tail_s for the reference's
    two-reduction order, k_pcg1_tail_s for the single-reduction one: the same iteration bodies, grid barriers in between).  Forced
    hand-over after `first` launched iterations: same iteration counts as the oracle, and BIT-IDENTICAL to the fully launched solve (tail
    and launched kernels share the virtual-workgroup grouping of the dot products: fixed 14 iterations, and a converging run that stops
    inside the tail)."""
    pos, vel, maxp = util.make_dam(*GRID)
    _, full = util.new_pair(*GRID, maxp)
    o, h = util.new_pair(*GRID, maxp)
    try:
        for f in (full, h):
            f.set_pcg_work_mapping("bricks")
            f.set_pcg_schedule(schedule)
        full.set_tuning("pcg_tail", 0)
        h.set_tuning("pcg_tail_first", first)
        o.set_particles(pos, *vel)
        run_until(o, "solve_velocity")
        util.copy_state(o, h)
        util.copy_state(o, full)
        state = {v: o.read_volume(v) for v in ("residual", "pressure_velocity", "search")}
        fluid = o.read_volume("marker") == 1
        for cfg in (dict(error_tolerance=0.0, max_num_iterations=14, error_check_frequency=4),
                    dict(error_tolerance=0.26, max_num_iterations=64, error_check_frequency=4)):
            for f in (o, h, full):
                f.set_solver_config(0, **cfg)
                for v, a in state.items():
                    f.write_volume(v, a)
            o.reset_pressure_cleared(0, False)
            h.mark_pressure_initialised(0, False)
            full.mark_pressure_initialised(0, False)
            o.run_stage("solve_velocity", util.DT)
            h.run_stage("solve_velocity", util.DT)
            full.run_stage("solve_velocity", util.DT)
            eo, io = o.solver_stats(0)
            eh, ih = h.solver_stats(0)
            ef, i_f = full.solver_stats(0)
            assert ih == io and ih >= 0, (cfg, (eh, ih), (eo, io))
            assert abs(eh - eo) <= (2e-2 if schedule == "single_reduction" else 2e-3) * eo      # (single reduction: a different rounding of the recurrence, tests/test_gpu_pcg_schedule.py)
            assert (ef, i_f) == (eh, ih), ((ef, i_f), (eh, ih))
            for name in ("pressure_velocity", "residual", "search"):
                a, b = h.read_volume(name), full.read_volume(name)
                assert np.array_equal(a[fluid], b[fluid]), name + ": tail and launched solve differ"
    finally:
        h.close()
        full.close()


@pytest.mark.parametrize("schedule", ["reference", "single_reduction"])
def test_pcg_results_do_not_depend_on_the_launch_grid(schedule):
    """The launch grid of the brick-mapped PCG kernels is an estimate from a lagged, asynchronous snapshot of the brick counts (round-2
    review: results moved with the host's timing).  The dot-product partials are grouped by VIRTUAL workgroups -- a function of the
    device-side brick list alone (pcg_vblocks, blub_pcg.hip.h) --, so the same solve launched with too few, about right and far too many
    workgroups gives bit-identical fields and statistics."""
    pos, vel, maxp = util.make_dam(*GRID)
    o, h = util.new_pair(*GRID, maxp)
    try:
        h.set_pcg_work_mapping("bricks")
        h.set_pcg_schedule(schedule)
        h.set_tuning("pcg_tail", 0)
        o.set_particles(pos, *vel)
        run_until(o, "solve_velocity")
        util.copy_state(o, h)
        state = {v: o.read_volume(v) for v in ("residual", "pressure_velocity", "search")}
        fluid = o.read_volume("marker") == 1
        h.set_solver_config(0, error_tolerance=0.05, max_num_iterations=40, error_check_frequency=4)
        results = []
        for grid in (0, 8, 24, 64, 1024):
            h.set_tuning("pcg_launch_grid", grid)
            for v, a in state.items():
                h.write_volume(v, a)
            h.mark_pressure_initialised(0, False)
            h.run_stage("solve_velocity", util.DT)
            results.append((h.solver_stats(0), h.read_volume("pressure_velocity")[fluid].copy(), h.read_volume("residual")[fluid].copy()))
        for stats, p, r in results[1:]:
            assert stats == results[0][0], (stats, results[0][0])
            assert np.array_equal(p, results[0][1]) and np.array_equal(r, results[0][2])
        assert 0 < results[0][0][1] <= 40
    finally:
        h.close()


def test_extrapolation_with_every_neighbour_count():
    """D3 on a random marker field (FLUID / AIR / SOLID mixed cell by cell): faces with 1 .. 8 valid in-plane neighbours all occur, so every
    entry of the kernel's constant-divisor table (k_extrapolate_b; n = 7 and 8 hardly ever appear in a dam-break scene) is compared
    bit for bit with the oracle's `avg / num`."""
    dim = (48, 32, 32)
    rng = np.random.default_rng(11)
    o, h = util.new_pair(*dim, 8)
    try:
        nz, ny, nx = dim[2], dim[1], dim[0]
        marker = rng.choice(np.array([1, -1, 0], np.int8), size=(nz, ny, nx), p=[0.45, 0.45, 0.10])
        marker[0] = marker[-1] = 0; marker[:, 0] = marker[:, -1] = 0; marker[:, :, 0] = marker[:, :, -1] = 0     # the domain shell is SOLID
        vel = {v: (rng.standard_normal((nz, ny, nx)) * 3).astype(np.float32) for v in ("vel_x", "vel_y", "vel_z")}
        for f in (o, h):
            f.write_volume("marker", marker)
            for v, a in vel.items():
                f.write_volume(v, a)
            f.write_volume("pressure_velocity", np.zeros((nz, ny, nx), np.float32))
        o.run_stage("project", util.DT)
        h.run_stage("project", util.DT)
        fl = marker == 1
        counts = set()
        for comp, (dz, dy, dx) in enumerate(((0, 0, 1), (0, 1, 0), (1, 0, 0))):
            a, b = h.read_volume(("vel_x", "vel_y", "vel_z")[comp]), o.read_volume(("vel_x", "vel_y", "vel_z")[comp])
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "component %d differs in %d cells" % (comp, (a != b).sum())
            # faces with a FLUID side, and per invalid face the number of valid in-plane neighbours (what the divisor table is indexed by)
            valid = fl | np.roll(fl, (-dz, -dy, -dx), (0, 1, 2))
            n = np.zeros(marker.shape, np.int32)
            axes = [ax for ax in range(3) if (dz, dy, dx)[ax] == 0]
            for s0 in (-1, 0, 1):
                for s1 in (-1, 0, 1):
                    if s0 or s1:
                        n += np.roll(np.roll(valid, s0, axes[0]), s1, axes[1])
            inner = np.zeros_like(fl); inner[2:-2, 2:-2, 2:-2] = True
            counts |= set(np.unique(n[inner & ~valid]).tolist())
        assert set(range(1, 9)) <= counts, counts
    finally:
        h.close()


@pytest.mark.parametrize("kind", ["single_full", "single_sparse", "dam", "dam_sparse", "single_full_odd_grid", "dam_odd_grid"])
def test_transfer_gather_kernels(kind):
    """The P2G gather exists twice: one lane per list cell of the tile (k_gather_velocity3_p) and with the tile's non-empty lists compacted
    (k_gather_velocity3_s; tiles with more than 256 lists take several passes, last slots first).  Both add a face's eight partial sums in
    the reference's list order.  "single_*": one particle per cell with fractional offsets in (0.5, 1) puts exactly one particle on every
    list of all three staggerings, so the lists do not depend on the order of the atomics and the velocity volumes of the two kernels must
    agree BIT FOR BIT (every tile cell occupied: three passes; a tenth of them: one pass).  "dam*": 8 particles per cell in random memory
    order -- the node order of a list is a race, so both kernels are held to the oracle at 1e-5 instead."""
    rng = np.random.default_rng(31)
    GRID = (44, 36, 30) if kind.endswith("odd_grid") else globals()["GRID"]      # partial bricks on all three upper faces (16 x 8 x 4 bricks)
    if kind.startswith("single"):
        nx, ny, nz = GRID
        cells = np.stack(np.meshgrid(np.arange(1, nx - 2), np.arange(1, int(ny * 0.7)), np.arange(1, nz - 2), indexing="ij"), -1).reshape(-1, 3)
        if kind == "single_sparse":
            cells = cells[rng.random(len(cells)) < 0.1]
        if kind.endswith("odd_grid"):           # up to the last cell layers: lists in the partial bricks
            cells = np.stack(np.meshgrid(np.arange(1, nx - 1), np.arange(1, ny - 1), np.arange(1, nz - 1), indexing="ij"), -1).reshape(-1, 3)
        pos = (cells + 0.5 + 0.49 * rng.random(cells.shape)).astype(np.float32)
        vel = []
        for c in range(3):
            rows = (rng.standard_normal((len(pos), 4)) * 0.3).astype(np.float32)
            rows[:, 3] = (3.0 * np.sin(pos[:, (c + 1) % 3] * 0.3 + c)).astype(np.float32)
            vel.append(rows)
        vel = tuple(vel)
        maxp = len(pos) + 64
    else:
        pos, vel, maxp = util.make_dam(*GRID, seed=5, fill=(0.97, 0.97, 1.0) if kind.endswith("odd_grid") else (0.45, 0.6, 1.0))
        if kind == "dam_sparse":
            keep = (rng.random(len(pos)) < 0.04) & (pos[:, 1] < 0.35 * GRID[1])
            pos, vel = pos[keep], tuple(v[keep] for v in vel)
    perm = rng.permutation(len(pos))
    pos, vel = pos[perm], tuple(v[perm] for v in vel)
    o, h = util.new_pair(*GRID, maxp)
    out = {}
    try:
        o.set_particles(pos, *vel)
        o.run_stage("transfer", util.DT)
        # the three gather kernels: tile-centric one lane per list cell (0), tile-centric compacted (1), list-centric + finishing kernel ("own")
        for key, (compact, own) in {0: (0, 0), 1: (1, 0), "own": (0, 1)}.items():
            h.set_tuning("p2g_compact", compact)
            h.set_tuning("p2g_own", own)
            h.set_particles(pos, *vel)
            h.run_stage("transfer", util.DT)
            out[key] = [h.read_volume(v) for v in ("vel_x", "vel_y", "vel_z")]
            assert np.array_equal(h.read_volume("marker"), o.read_volume("marker"))
            for a, name in zip(out[key], ("vel_x", "vel_y", "vel_z")):
                util.assert_close(name, a, o.read_volume(name), rel=1e-5)
                assert np.abs(a).max() > 0.5
        if kind.startswith("single"):
            for a, b, name in zip(out[0], out[1], "xyz"):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "vel_%s differs in %d cells" % (name, (a != b).sum())
            # list-centric: the same particle-face products; faces on a brick's negative sides add them as per-brick subtotals (another association of
            # <= 8 partial sums), every other face bit for bit
            for a, b, name in zip(out[0], out["own"], "xyz"):
                differ = a != b
                x, y, z = np.nonzero(differ.transpose(2, 1, 0))      # (volumes are [z, y, x])
                assert np.all((x % 16 == 0) | (y % 8 == 0) | (z % 4 == 0)), "vel_%s: a face inside a brick differs" % name
                assert np.abs(a - b).max() <= 4e-6 * max(1.0, np.abs(a).max()), np.abs(a - b).max()
                print("list-centric vs tile-centric vel_%s: %d of %d written faces differ in the last bits (all on brick boundaries)" % (name, differ.sum(), (a != 0).sum()))
    finally:
        h.close()


def test_transfer_with_shuffled_particle_order():
    """P2G with the particles in RANDOM memory order (what the order decays to between two rebinnings): the lanes of a wave that insert
    into the same list are no longer adjacent, so this exercises the wave-wide grouping of wave_list_insert (one atomic per distinct key).
    The three gathers must agree with the oracle to 1e-5 -- every list member contributes to eight faces, so a lost, duplicated or
    misplaced list node shows -- and every particle must be on exactly one x list (the engine keeps the three staggered lists at once,
    the oracle re-uses one volume per component like the reference: the list volumes themselves are not comparable after the stage)."""
    pos, vel, maxp = util.make_dam(*GRID)
    rng = np.random.default_rng(21)
    perm = rng.permutation(len(pos))
    pos = pos[perm]
    vel = tuple(v[perm] for v in vel)
    o, h = util.new_pair(*GRID, maxp)
    try:
        o.set_particles(pos, *vel)
        h.set_particles(pos, *vel)
        o.run_stage("transfer", util.DT)
        h.run_stage("transfer", util.DT)
        assert np.array_equal(h.read_volume("marker"), o.read_volume("marker"))
        n = len(pos)
        lh = util.lists_as_sets(h.read_volume("linked_list"), h.get_particles()[0], n)
        members = np.concatenate([np.fromiter(s, np.int64) for s in lh.values()])
        assert len(members) == n and len(np.unique(members)) == n            # a partition of the particles
        dual = np.floor(pos - np.array([1.0, 0.5, 0.5], np.float32)).astype(np.int64)
        cell_of = (dual[:, 2] * GRID[1] + dual[:, 1]) * GRID[0] + dual[:, 0]
        for cell, s in list(lh.items())[::97]:
            assert all(cell_of[i] == cell for i in s)
        assert max(len(s) for s in lh.values()) <= 12
        for v in ("vel_x", "vel_y", "vel_z"):
            util.assert_close(v, h.read_volume(v), o.read_volume(v), rel=1e-5)
    finally:
        h.close()
