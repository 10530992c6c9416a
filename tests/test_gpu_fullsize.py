"""BASELINE.json configurations at their FULL sizes on the GPU: size-independent properties (the oracle is only used where it
finishes in seconds).  configs[1] dam_halfhalf 128x64x64 / 1.2 M particles, configs[2] double_dam, the headline
corner_dams_256 (968 688 particles @ 256^3), configs[3] dam_halfhalf_highres 256x128x128 / 10.1 M particles."""
import os

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _scene(name):
    import blub_amd
    return blub_amd.Scene(path=os.path.join(ROOT, "scenes", name + ".json"))


def _records(a):
    return np.sort(np.ascontiguousarray(a[:, :3]).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=("x", "y", "z"))


@pytest.mark.parametrize("name,particles", [("dam_halfhalf", 1218672), ("double_dam", 1199328), ("corner_dams_256", 968688)])
def test_full_scene_invariants(name, particles):
    scene = _scene(name)
    f = scene.fluid()
    try:
        nx, ny, nz = f.grid_dimension()
        assert f.num_particles() == particles
        p0 = f.get_particles()[0]
        for _ in range(12):
            scene.step(util.DT)
        f.synchronize()
        pos, vx, vy, vz = f.get_particles()
        assert pos.shape[0] == particles                                           # nothing lost
        assert np.all(np.isfinite(pos[:, :3])) and all(np.all(np.isfinite(v)) for v in (vx, vy, vz))
        lo, hi = np.float32(1.001), np.array([nx, ny, nz], np.float32) - np.float32(1.001)
        assert np.all(pos[:, :3] >= lo) and np.all(pos[:, :3] <= hi)             # advect_particles.comp:167 clamp
        # step 0 rebinned (Q13): the particle set is a permutation of the seeded one advanced by 12 steps: same count per
        # x-z column is NOT conserved, but gravity must have moved the centre of mass down and nothing sideways on average
        assert pos[:, 1].mean() < p0[:, 1].mean()
        # marker == cells holding a particle (+ SOLID shell), exactly, after the last advect... the density correction
        # moved particles afterwards, so compare against the marker the NEXT transfer builds:
        f.run_stage("transfer", util.DT)
        marker = f.read_volume("marker")
        cells = pos[:, :3].astype(np.int64)
        expect = -np.ones((nz, ny, nx), np.int8)
        expect[[0, -1], :, :] = 0; expect[:, [0, -1], :] = 0; expect[:, :, [0, -1]] = 0
        expect[cells[:, 2], cells[:, 1], cells[:, 0]] = 1
        assert np.array_equal(marker, expect)
        # velocities written by P2G are finite and zero far away from the fluid
        v = f.read_volume("vel_y")
        assert np.all(np.isfinite(v)) and np.abs(v).max() > 0
        # solver statistics: iteration counts obey the check cadence, errors are positive and bounded
        for stats in (f.pressure_solver_stats_velocity(), f.pressure_solver_stats_density()):
            assert len(stats) == 12
            for s in stats:
                assert s.iteration_count == 32 or (s.iteration_count > 0 and s.iteration_count % 4 == 0)
                assert 0 <= s.error < 50 and (s.iteration_count == 32 or s.error < 0.1)
        bc = f.brick_counts()
        assert 0 < bc["fluid"] <= bc["active"] <= bc["total"]
    finally:
        f.close()


def test_full_size_binning_is_a_permutation():
    """configs[3] scale: 10.1 M particles, 256x128x128: prefix-sum binning (count / wave-shuffle scan / rewrite)."""
    scene = _scene("dam_halfhalf_highres")
    f = scene.fluid()
    try:
        assert f.num_particles() == 10113264
        before = f.get_particles()[0]
        rng = np.random.default_rng(0)
        perm = rng.permutation(len(before))
        f.set_particles(before[perm])
        f.run_stage("binning", util.DT)
        after = f.get_particles()[0]
        assert np.array_equal(_records(after), _records(before))
        c = after[:, :3].astype(np.int64)
        nx, ny, nz = f.grid_dimension()
        assert np.all(np.diff((c[:, 2] * ny + c[:, 1]) * nx + c[:, 0]) >= 0)
        f.step(util.DT)          # and one full step at this size runs
        f.synchronize()
        assert f.pressure_solver_stats_velocity()[-1].iteration_count in range(4, 33)
    finally:
        f.close()


@pytest.mark.parametrize("n,mapping", [(256, "rows"), (128, "bricks"), (128, "bricks_single")])
def test_pcg_full_grid_solution_satisfies_the_linear_system(n, mapping):
    """Dense n^3 Poisson problem (SOLID shell, FLUID inside, one AIR layer under the lid so that the system is not the
    singular pure-Neumann one; 16.3 M unknowns at 256^3): after the solve the engine's own residual volume must equal
    b - A p recomputed on the host in f64, and the reported error must equal max|r| * dt."""
    import blub_amd
    h = blub_amd.HybridFluid((n, n, n), 16, binning="off")
    try:
        util.set_mapping(h, mapping)
        marker = np.zeros((n, n, n), np.int8)
        marker[1:-1, 1:-1, 1:-1] = 1
        marker[1:-1, -2, 1:-1] = -1          # AIR: Dirichlet p = 0
        ax = np.sin(2 * np.pi * (np.arange(n) + 0.5) / n).astype(np.float32)
        b = (ax[:, None, None] * ax[None, :, None] * ax[None, None, :]).astype(np.float32)
        b[marker != 1] = 0
        h.write_volume("marker", marker)
        h.write_volume("residual", b)
        h.set_solver_config(0, error_tolerance=0.0, max_num_iterations=24, error_check_frequency=4)
        h.run_stage("solve_velocity", util.DT)
        p = h.read_volume("pressure_velocity").astype(np.float64)
        r = h.read_volume("residual").astype(np.float64)
        err, it = h.solver_stats(0)
        assert it == 24
        fluid = marker == 1
        assert np.all(p[~fluid] == 0)
        Ap = np.zeros_like(p)
        inner = (slice(1, -1),) * 3
        diag = np.zeros_like(p)
        for axis in range(3):
            for sft in (-1, 1):
                nb_fluid = np.roll(fluid, sft, axis)
                Ap -= np.roll(p, sft, axis) * nb_fluid
                diag += np.roll(marker != 0, sft, axis)          # diagonal = number of non-SOLID neighbours
        Ap = (Ap + diag * p) * fluid
        r_expected = (b - Ap) * fluid
        # the recurrence r -= alpha A s drifts from b - A p by O(eps * |A| |p|) per iteration in f32
        assert np.abs(r * fluid - r_expected).max() < 3e-6 * (12 * np.abs(p).max() + np.abs(b).max())
        assert abs(np.abs(r[fluid]).max() * util.DT - err) <= 1e-5 * err + 1e-12
        # (no monotonicity claim: with this smooth right-hand side and the z = r/d^2 preconditioner max|r| first grows --
        #  the CPU oracle shows 217 / 75 / 94 / 77 after 4 / 8 / 16 / 24 iterations at 128^3 -- CG minimises the A-norm of the error)
        if n == 128:
            # max|r| of this UNCONVERGED iterate is a rounding amplifier (the oracle reports 76.57 with f64 dot products and 92.60 with f32 ones, the
            # engine's f32 trees 71.5 - 73.7): an envelope around it says nothing (round-4 review, item 4d: the +-20 assertion that stood here).  What CG
            # minimises is the energy phi(p) = p.Ap / 2 - b.p (the A-norm of the error up to a constant), and THAT is insensitive to the rounding of the
            # dots: the engine's iterate must reach the energy of the oracle's 24th iterate (f64 dots) to 5e-4 (measured: < 1e-4 with the reference's
            # order of operations, 1.8e-4 with the single-reduction recurrence) -- a solver that lost an iteration, dropped a neighbour term or used a
            # wrong alpha anywhere would miss it by percents.
            from oracle.oracle import Oracle
            o = Oracle(n, n, n, 16)
            o.set_quirks(precond="zero", binning="off")
            o.write_volume("marker", marker)
            o.write_volume("residual", b)
            o.write_volume("pressure_velocity", np.zeros_like(b))
            o.reset_pressure_cleared(0, True)
            o.set_solver_config(0, error_tolerance=0.0, max_num_iterations=24, error_check_frequency=4)
            o.run_stage("solve_velocity", util.DT)
            po = o.read_volume("pressure_velocity").astype(np.float64)

            def energy(q):
                Aq = np.zeros_like(q)
                for axis in range(3):
                    for sft in (-1, 1):
                        Aq -= np.roll(q, sft, axis) * np.roll(fluid, sft, axis)
                Aq = (Aq + diag * q) * fluid
                return 0.5 * float((q * Aq).sum()) - float((b.astype(np.float64) * q).sum())
            e_engine, e_oracle = energy(p), energy(po)
            print("energy after 24 iterations, %s: engine %.8g oracle %.8g (relative difference %.3g)" % (mapping, e_engine, e_oracle, abs(e_engine - e_oracle) / abs(e_oracle)))
            assert e_oracle < 0 and abs(e_engine - e_oracle) <= 5e-4 * abs(e_oracle), (mapping, e_engine, e_oracle)
        # linearity: solving 2b from the same start gives 2p (CG is scale invariant)
        h.write_volume("residual", 2 * b)
        h.mark_pressure_initialised(0, False)
        h.run_stage("solve_velocity", util.DT)
        p2 = h.read_volume("pressure_velocity").astype(np.float64)
        assert np.abs(p2 - 2 * p).max() <= 2e-4 * np.abs(p).max()
    finally:
        h.close()


def test_sixty_steps_of_the_reference_shaped_scene_track_the_oracle():
    """corner_dams_128 (the BASELINE scene family at 128^3, 111 600 particles) with the reference's defaults -- tolerance 0.1,
    32 iterations, check every 4, rebinning at step 0 -- for 60 steps on both sides.  Individual particles diverge (chaotic
    system, loosely converged solves), so the comparison is statistical; the first steps must agree closely.
    Measured: steps 1-2 identical iteration counts and errors (step 3 within a few %); step 60: centre of mass 7e-3 cells apart, kinetic energy 0.4 %,
    occupancy-histogram L1 0.22."""
    import os
    import blub_amd
    from oracle.oracle import Oracle
    from tests.conftest import ROOT
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_128.json"))
    f = scene.fluid()
    try:
        dim = f.grid_dimension()
        o = Oracle(dim[0], dim[1], dim[2], f.num_particles() + 64)
        o.set_particles(f.get_particles()[0])
        o.set_gravity_grid(np.float32(list(scene.config.gravity)) / np.float32(scene.config.grid_to_world_scale))
        occ = lambda p: np.bincount(((p[:, 2].astype(int) * dim[1] + p[:, 1].astype(int)) * dim[0] + p[:, 0].astype(int)), minlength=int(np.prod(dim)))
        for step in range(1, 61):
            scene.step(util.DT)
            o.step(util.DT)
            if step <= 3:
                f.synchronize()
                f.update_statistics()
                for w, hist in ((0, f.pressure_solver_stats_velocity()), (1, f.pressure_solver_stats_density())):
                    err_o, it_o = o.solver_stats(w)
                    if step <= 2:   # before the two sides' rounding has had time to spread
                        assert hist[-1].iteration_count == it_o, (step, w, hist[-1], it_o)
                        assert abs(hist[-1].error - err_o) <= 0.03 * err_o, (step, w, hist[-1], err_o)
                    else:           # (run to run the GPU's own step-3 error moves by 2-4 %: atomic list order)
                        assert abs(hist[-1].iteration_count - it_o) <= 4 and abs(hist[-1].error - err_o) <= 0.15 * err_o, (step, w, hist[-1], it_o, err_o)
        pg, po = f.get_particles(), o.get_particles()
        a, b = pg[0][:, :3].astype(np.float64), po[0][:, :3].astype(np.float64)
        assert a.shape == b.shape
        ke = lambda q: sum((q[c][:, 3].astype(np.float64) ** 2).sum() for c in (1, 2, 3))
        com, l1, ker = np.abs(a.mean(0) - b.mean(0)).max(), np.abs(occ(a) - occ(b)).sum() / len(a), ke(pg) / ke(po)
        print("step 60: centre of mass %.3g cells, occupancy L1 %.3g, kinetic energy ratio %.4f" % (com, l1, ker))
        assert com < 0.05 and l1 < 0.5 and abs(ker - 1.0) < 0.03
    finally:
        f.close()


@pytest.mark.parametrize("schedule", ["reference", "single_reduction"])
def test_brick_mapped_solve_is_bit_reproducible_across_launch_grids(schedule):
    """A brick-mapped solve groups its dot-product partials by virtual workgroups that depend on the brick list alone (pcg_vblocks,
    blub_pcg.hip.h): on a grid with partial bricks at its upper faces and a ragged fluid region, repeated solves and solves launched
    with very different grids give the same bits."""
    import blub_amd
    dim = (72, 44, 30)
    rng = np.random.default_rng(11)
    marker = -np.ones(dim[::-1], np.int8)
    marker[[0, -1], :, :] = 0; marker[:, [0, -1], :] = 0; marker[:, :, [0, -1]] = 0
    blob = rng.random(dim[::-1]) < 0.55
    blob[:, 30:, :] = False
    marker[(marker == -1) & blob] = 1
    marker[10:14, 5:9, 20:30] = 0
    b = np.where(marker == 1, rng.standard_normal(dim[::-1]), 0).astype(np.float32)
    out = {}
    for grid in (0, 16, 512, -1):          # -1: the estimate again (run-to-run)
        h = blub_amd.HybridFluid(dim, 8, binning="off")
        try:
            h.set_pcg_work_mapping("bricks")
            h.set_pcg_schedule(schedule)
            h.set_tuning("pcg_launch_grid", max(grid, 0))
            h.write_volume("marker", marker)
            h.write_volume("residual", b)
            h.set_solver_config(0, error_tolerance=0.0, max_num_iterations=9, error_check_frequency=4)
            h.run_stage("solve_velocity", util.DT)
            out[grid] = (h.read_volume("pressure_velocity"), h.read_volume("residual"), h.read_volume("search"), h.solver_stats(0))
        finally:
            h.close()
    a = out[0]
    assert np.abs(a[0]).max() > 0
    fluid = marker == 1
    for grid in (16, 512, -1):
        c = out[grid]
        for k in range(3):
            assert np.array_equal(a[k][fluid].view(np.uint32), c[k][fluid].view(np.uint32)), (grid, k)
        assert a[3] == c[3]
