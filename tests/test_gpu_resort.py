"""The engine's own particle re-sort (round 6; blub_amd/csrc/blub_bricks.hip.h "internal re-sort", include/blubhip.h: "resort_every").

The reference only reorders particles when it rebins (every 60 steps, hybrid_fluid.rs:854-893); the engine re-sorts its arrays by (brick, cell)
every few steps because every particle kernel follows the order of the particles in memory.  What the CALLER sees must not change: the
order of blub_fluid_get_particles, the particle indices stored in the list links and heads, the reference's rebinning cadence."""
import numpy as np
import pytest

from tests import util
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

GRID = (64, 48, 32)
CONVERGED = dict(error_tolerance=2e-6, max_num_iterations=400, error_check_frequency=8)


def _engine(resort_every, binning="off"):
    import blub_amd
    pos, vel, maxp = util.make_dam(*GRID, seed=3)
    h = blub_amd.HybridFluid(GRID, maxp, binning=binning)
    h.set_gravity_grid((0.0, -981.0, 0.0))
    h.set_tuning("resort_every", resort_every)
    h.set_particles(pos, *vel)
    for w in (0, 1):
        h.set_solver_config(w, **CONVERGED)
    return h, pos, vel


def test_resorting_every_second_step_leaves_the_callers_particle_order_alone():
    """Rebinning off (the caller's order must never change), converged solves (nothing amplifies the order of the list atomics), six steps: an engine that
    re-sorts at steps 2 and 4 returns THE SAME particles under the same indices as one that never re-sorts -- positions and the three APIC rows,
    particle by particle (the internal order changes which lanes share a list atomic, i.e. the rounding of the gathers: 1e-4 cells, as for any two
    runs) -- and the lists it shows are lists of the caller's indices: every particle in exactly the density list of its own dual cell."""
    a, pos, vel = _engine(2)
    b, _, _ = _engine(0)
    try:
        for _ in range(6):
            a.step(util.DT)
            b.step(util.DT)
        pa, pb = a.get_particles(), b.get_particles()
        d = np.abs(pa[0][:, :3] - pb[0][:, :3]).max(axis=1)
        moved = np.abs(pb[0][:, :3] - pos).max(axis=1)
        print("re-sorted vs never re-sorted after 6 steps: median %.3g p99.9 %.3g max %.3g cells (the particles moved %.3g cells on average)" % (
            np.median(d), np.quantile(d, 0.999), d.max(), moved.mean()))
        assert moved.mean() > 0.1
        assert np.median(d) < 2e-5 and np.quantile(d, 0.999) < 5e-4 and util.max_but_three(d) < 2e-2 and d.max() < 1.0      # (measured 4.8e-6 / 1.2e-4 / 2e-3; a permutation would show as tens of cells)
        for c in (1, 2, 3):
            dv = np.abs(pa[c] - pb[c]).max(axis=1)
            assert np.quantile(dv, 0.999) < 1e-2 * max(1.0, np.abs(pb[c]).max()), (c, np.quantile(dv, 0.999))
        # the density lists of the last step (heads volume + links in particles_position_ll.w), in the CALLER's indices
        n = len(pos)
        heads = a.read_volume("linked_list")
        lists = util.lists_as_sets(heads, a.get_particles()[0], n)
        seen = np.zeros(n, np.int32)
        nx, ny, nz = GRID
        # (the lists were built by the advection, before the density correction moved the particles once more: membership is checked by COUNT and
        #  by neighbourhood -- every particle in exactly one list, within 2 cells of that list's dual cell)
        final = a.get_particles()[0][:, :3]
        for cell, members in lists.items():
            idx = np.fromiter(members, np.int64)
            seen[idx] += 1
            cz, cy, cx = cell // (nx * ny), (cell // nx) % ny, cell % nx
            assert np.all(np.abs(final[idx] - 0.5 - np.array([cx, cy, cz]) - 0.5).max(axis=1) < 2.5), cell
        assert np.all(seen == 1), "every particle belongs to exactly one density list (%d do not)" % (seen != 1).sum()
        # reading twice changes nothing; stepping on after the order was restored works
        again = a.get_particles()
        assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a.get_particles(), again))      # (bit patterns: a list end is 0xFFFFFFFF, a NaN)
        a.step(util.DT); b.step(util.DT)
        d7 = np.abs(a.get_particles()[0][:, :3] - b.get_particles()[0][:, :3]).max(axis=1)
        assert np.quantile(d7, 0.999) < 1e-3
    finally:
        a.close()
        b.close()


def test_the_references_rebinning_cadence_defines_the_callers_order_also_between_resorts():
    """Rebinning "fixed" every 4 steps, re-sorting every 2: after a rebinning step the caller's order is the reference's cell order (linear index, x
    fastest: particle_binning_prefixsum.comp:17-22), and it STAYS that order -- the same particle under the same index -- over the engine's own
    re-sort two steps later."""
    import blub_amd
    a, pos, vel = _engine(2, binning="fixed")
    try:
        a.particle_rebinning_step_frequency = 4
        for _ in range(5):      # steps 0 .. 4: rebinned in steps 0 and 4
            a.step(util.DT)
        p5 = a.get_particles()[0][:, :3].copy()
        # the order the reference's rebinning of step 4 left (positions at that time), then advected and corrected once: still nearly cell-sorted
        cell = np.floor(p5).astype(np.int64)
        lin = (cell[:, 2] * GRID[1] + cell[:, 1]) * GRID[0] + cell[:, 0]
        assert (np.diff(lin) < 0).mean() < 0.2
        a.step(util.DT)         # step 5
        p6 = a.get_particles()[0][:, :3].copy()
        a.step(util.DT)         # step 6: the engine re-sorts (6 % 2 == 0, no rebinning)
        p7 = a.get_particles()[0][:, :3]
        # the same particle under the same index: a step moves a particle by less than a cell or two, a permutation would move most indices across the domain
        assert np.abs(p6 - p5).max() < 3.0 and np.abs(p7 - p6).max() < 3.0
    finally:
        a.close()


def test_entry_points_that_take_particles_by_index_see_the_callers_order():
    """set_particles / add_fluid_cube / the stage hook after re-sorted steps: the engine restores the caller's order first, so a stage on the restored
    state equals the same stage on a fresh engine that was handed the same arrays."""
    import blub_amd
    a, pos, vel = _engine(1)
    try:
        for _ in range(3):      # re-sorted in steps 1 and 2
            a.step(util.DT)
        st = a.get_particles()
        b = blub_amd.HybridFluid(GRID, len(pos) + 64, binning="off")
        try:
            b.set_gravity_grid((0.0, -981.0, 0.0))
            b.set_particles(*st)
            for f in (a, b):
                f.run_stage("transfer", util.DT)
            # (which 12 particles a list beyond the cap keeps is the order of the list atomics between waves -- a race between ANY two runs: the faces such
            #  lists reach are left out, tests/test_gpu_baseline_parity.py)
            from tests.test_gpu_baseline_parity import _faces_reached_by_long_lists
            reach = _faces_reached_by_long_lists(st[0], GRID)
            for c, v in enumerate(("vel_x", "vel_y", "vel_z")):
                va, vb = a.read_volume(v), b.read_volume(v)
                keep = (vb != 0) & ~reach[c][0]
                assert keep.sum() > 10000
                util.assert_close_but_few(v, np.where(keep, va, 0), np.where(keep, vb, 0), rel=1e-5)      # (two engines, two particle orders: the order of a face's additions)
            assert np.array_equal(a.read_volume("marker"), b.read_volume("marker"))
            assert np.array_equal(a.read_volume("linked_list"), b.read_volume("linked_list")) or util.lists_as_sets(
                a.read_volume("linked_list"), a.get_particles()[0], len(pos)) == util.lists_as_sets(b.read_volume("linked_list"), b.get_particles()[0], len(pos))
        finally:
            b.close()
    finally:
        a.close()
