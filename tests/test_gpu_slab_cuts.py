"""z-slab groups with the caller's own cut planes (round-4 review, item 1a: fluid-weighted cuts -- six of eight uniform slabs of the metric's scene
own no fluid), and the two round-4 ADVICE items about slab geometry:

  * planes whose byte count is not a multiple of 16 (the 1-byte descriptor plane of a 20 x 30 grid, the 4-byte count gather) used to leave the
    batched push of the direct transport -- an early flag raise plus a plain copy into the peer's memory; they now travel inside the one push kernel;
  * a slab whose plane COUNT equals nz while its first held plane is not 0 (every rank allocates the thickest slab's count) took the whole-grid path
    of read_volume / write_volume and touched memory in front of its allocation.
"""
import os

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _blob(dim, lo, hi, seed=4):
    rng = np.random.default_rng(seed)
    cells = np.stack(np.meshgrid(np.arange(lo[0], hi[0]), np.arange(lo[1], hi[1]), np.arange(lo[2], hi[2]), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    vel = [np.zeros((pos.shape[0], 4), np.float32) for _ in range(3)]
    vel[2][:, 3] = 6.0 * np.sin(pos[:, 0] * 0.4)            # z-velocities push particles across the interfaces
    return pos, vel


def _group_tracks_single(dim, pos, vel, slabs, transport, cuts, steps=3):
    """The envelope of tests/test_gpu_parity.py::test_z_slab_decomposition_matches_single_domain (same bounds, same reasons)."""
    import blub_amd
    from tests.test_gpu_parity import _match_particles
    cfg = dict(error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=slabs, binning="off", cuts=cuts)
    group.set_transport(transport)
    for f in [single] + [group.local_fluid(i) for i in range(slabs)]:
        f.set_tuning("pcg1_max_iterations", 1000)
    try:
        for f in (single, group):
            f.set_gravity_grid((0.0, -981.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        ranges = [group.local_range(i) for i in range(slabs)]
        if cuts is not None:
            assert group.cuts() == list(cuts) and ranges == list(zip(cuts[:-1], cuts[1:]))
        assert ranges[0][0] == 0 and ranges[-1][1] == dim[2] and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        counts0 = [group.local_fluid(i).num_particles() for i in range(slabs)]
        assert sum(counts0) == pos.shape[0]
        for step in range(steps):
            single.step(util.DT)
            group.step(util.DT)
            ps = single.get_particles()[0][:, :3].astype(np.float64)
            pg = group.get_particles()[0][:, :3].astype(np.float64)
            assert pg.shape == ps.shape
            d = _match_particles(pg, ps)
            q = (np.median(d), np.quantile(d, 0.99), np.quantile(d, 0.999), d.max())
            print("step %d  %d slabs %s cuts %s: median %.3g p99 %.3g p99.9 %.3g max %.3g" % ((step, slabs, transport, cuts) + q))
            bounds = (3e-5, 4e-4, 1.5e-3, 3e-3) if step == 0 else (2e-4, 3e-3, 3e-2, 0.1)
            for a, b in zip(q, bounds):
                assert a <= b, (step, q, bounds)
        counts1 = [group.local_fluid(i).num_particles() for i in range(slabs)]
        assert sum(counts1) == pos.shape[0] and counts1 != counts0
        pgl = group.get_particles()[0]
        off = 0
        for i, (z0, z1) in enumerate(ranges):
            z = pgl[off:off + counts1[i], 2]
            off += counts1[i]
            assert np.all(z >= z0) and np.all(z < z1)
        for w in (0, 1):
            st = [group.local_fluid(i).solver_stats(w) for i in range(slabs)]
            assert all(x == st[0] for x in st)
        if transport == "direct":
            assert group.host_syncs() == (0, 0) and group.held_back() == 0
        return counts0
    finally:
        single.close()
        group.close()


@pytest.mark.parametrize("transport", ["direct", "host"])
@pytest.mark.parametrize("cuts", [(0, 8, 12, 28, 48), (0, 4, 40, 44, 48), "balanced"])
def test_a_group_with_its_own_cut_planes_matches_the_single_domain(cuts, transport):
    """Uneven slabs -- one brick layer thin next to nine layers thick, and the balanced cuts of the blob itself --: every exchange (ghost particles,
    halo planes, tagged partials, migration) addresses its neighbour through the per-rank first plane, so nothing may assume equal slabs."""
    import blub_amd
    dim = (32, 32, 48)
    pos, vel = _blob(dim, (6, 8, 6), (26, 20, 42))
    if cuts == "balanced":
        cuts, bricks = blub_amd.SlabGroup.balanced_cuts(dim, pos, 4)
        assert max(bricks) <= 1.35 * (sum(bricks) / 4.0), (cuts, bricks)
        cuts = tuple(cuts)
    _group_tracks_single(dim, pos, vel, len(cuts) - 1, transport, cuts)


@pytest.mark.parametrize("transport", ["direct", "host"])
def test_planes_that_are_not_a_multiple_of_16_bytes(transport):
    """20 x 30 cells per plane: the descriptor plane is 600 bytes (not a multiple of 16; its address is not 16-byte aligned either on odd planes), the
    count gather 4 bytes.  Round-4 ADVICE (medium): these used to flush the batch mid-exchange and go through hipMemcpyAsync."""
    dim = (20, 30, 48)
    pos, vel = _blob(dim, (3, 4, 6), (17, 18, 42), seed=6)
    _group_tracks_single(dim, pos, vel, 3, transport, None)


def test_read_and_write_volume_of_a_slab_whose_plane_count_equals_nz():
    """nz = 24 as 3 slabs: [0, 8), [8, 16), [16, 24).  The middle slab holds [0, 24) -- 24 planes --, and every rank allocates that count; slab 2's first held
    plane is 8, so its plane count equals nz while its pointers are allocation - 8 planes.  Round-4 ADVICE (medium): read / write_volume took the
    whole-grid path and touched 8 planes in front of the allocation (the neighbouring volume of the one device allocation)."""
    import blub_amd
    dim = (32, 32, 24)
    group = blub_amd.SlabGroup(dim, 1024, local=3)
    try:
        f = group.local_fluid(2)
        z0, z1 = group.local_range(2)
        assert (z0, z1) == (16, 24)
        held = slice(z0 - 8, 24)
        before = {v: f.read_volume(v) for v in ("vel_x", "vel_y", "pressure_velocity", "marker")}
        assert np.all(before["marker"][:z0 - 8] == 0) and np.all(before["marker"][z0 - 8:23, 1:-1, 1:-1] == -1) and np.all(before["marker"][23] == 0)      # (plane 23: the SOLID shell)
        v = np.zeros(dim[::-1], np.float32)
        v[:] = 1.0 + np.arange(24, dtype=np.float32)[:, None, None]
        f.write_volume("vel_y", v)
        back = f.read_volume("vel_y")
        assert np.array_equal(back[held], v[held]) and not back[:z0 - 8].any()
        for name in ("vel_x", "pressure_velocity", "marker"):      # the neighbouring volumes of the allocation are untouched
            assert np.array_equal(f.read_volume(name), before[name]), name
        # and the same through every slab of the group: each sees exactly its held planes
        for i in range(3):
            fi = group.local_fluid(i)
            a, b = group.local_range(i)
            fi.write_volume("vel_x", v)
            r = fi.read_volume("vel_x")
            lo = max(0, a - 8)
            hi = min(24, lo + 24)      # (every slab allocates the plane count of the thickest one: 24 here, from its own first held plane on)
            assert np.array_equal(r[lo:hi], v[lo:hi]) and not r[:lo].any() and not r[hi:].any()
    finally:
        group.close()


@pytest.mark.parametrize("transport", ["direct", "host"])
def test_moving_the_cut_planes_of_a_running_group(transport):
    """blub_slab_group_recut (round 5): four slabs that hold the whole grid (movable_cuts), two steps on (0, 12, 24, 36, 48), the cut planes move to
    (0, 8, 28, 40, 48) -- pressure planes and particles change owner between adjacent slabs --, two more steps: the group stays on the single domain's
    trajectory, every slab holds exactly the particles of its NEW range, no particle is lost, and a cut that would jump over its neighbour is refused."""
    import blub_amd
    from tests.test_gpu_parity import _match_particles
    dim = (32, 32, 48)
    pos, vel = _blob(dim, (6, 8, 6), (26, 20, 42))
    cfg = dict(error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=4, binning="off", cuts=(0, 12, 24, 36, 48), movable_cuts=True)
    group.set_transport(transport)
    for f in [single] + [group.local_fluid(i) for i in range(4)]:
        f.set_tuning("pcg1_max_iterations", 1000)
    try:
        for f in (single, group):
            f.set_gravity_grid((0.0, -981.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        for step in range(4):
            if step == 2:
                with pytest.raises(blub_amd.BlubError):
                    group.recut((0, 24, 28, 40, 48))          # cut 1 would land ON old cut 2
                before = group.num_particles()
                group.recut((0, 8, 28, 40, 48))
                assert group.cuts() == [0, 8, 28, 40, 48] and [group.local_range(i) for i in range(4)] == [(0, 8), (8, 28), (28, 40), (40, 48)]
                assert group.num_particles() == before == pos.shape[0]
            single.step(util.DT)
            group.step(util.DT)
            ps = single.get_particles()[0][:, :3].astype(np.float64)
            pgl = group.get_particles()[0]
            d = _match_particles(pgl[:, :3].astype(np.float64), ps)
            q = (np.median(d), np.quantile(d, 0.99), np.quantile(d, 0.999), d.max())
            print("step %d (%s, cuts %s): median %.3g p99 %.3g p99.9 %.3g max %.3g" % ((step, transport, group.cuts()) + q))
            bounds = (3e-5, 4e-4, 1.5e-3, 3e-3) if step == 0 else (2e-4, 3e-3, 3e-2, 0.1)
            for a, b in zip(q, bounds):
                assert a <= b, (step, q, bounds)
            off = 0
            for i in range(4):
                z0, z1 = group.local_range(i)
                n = group.local_fluid(i).num_particles()
                z = pgl[off:off + n, 2]
                off += n
                assert np.all(z >= z0) and np.all(z < z1), (step, i)
        for w in (0, 1):
            st = [group.local_fluid(i).solver_stats(w) for i in range(4)]
            assert all(x == st[0] for x in st)
    finally:
        single.close()
        group.close()


def test_random_legal_recuts_keep_the_trajectory():
    """Ten steps with a RANDOM legal re-cut in front of every step (each cut strictly between its old neighbours, slabs down to one brick layer, slabs
    that own no particle at all): particle count conserved, ownership exact after every re-cut, and the group stays inside the free-running envelope of the
    single domain -- a re-cut moves owners, never the physics."""
    import blub_amd
    from tests.test_gpu_parity import _match_particles
    dim = (32, 32, 48)
    pos, vel = _blob(dim, (6, 8, 6), (26, 20, 42))
    cfg = dict(error_tolerance=0.0, max_num_iterations=60, error_check_frequency=8)
    rng = np.random.default_rng(2025)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=5, binning="off", movable_cuts=True)
    try:
        for f in (single, group):
            f.set_gravity_grid((0.0, -981.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        seen = set()
        for step in range(10):
            old = group.cuts()
            new = list(old)
            for r in range(1, 5):      # a new cut anywhere strictly between the OLD neighbours, and above the new cut below it
                lo, hi = max(old[r - 1], new[r - 1]) + 4, old[r + 1] - 4
                if lo <= hi:
                    new[r] = int(rng.integers(lo // 4, hi // 4 + 1)) * 4
            group.recut(new)
            assert group.cuts() == new
            seen.add(tuple(new))
            single.step(util.DT)
            group.step(util.DT)
            assert group.num_particles() == pos.shape[0]
            pgl = group.get_particles()[0]
            off = 0
            for i in range(5):
                z0, z1 = group.local_range(i)
                n = group.local_fluid(i).num_particles()
                z = pgl[off:off + n, 2]
                off += n
                assert np.all(z >= z0) and np.all(z < z1), (step, i, (z0, z1))
        d = _match_particles(pgl[:, :3].astype(np.float64), single.get_particles()[0][:, :3].astype(np.float64))
        q = (np.median(d), np.quantile(d, 0.99), np.quantile(d, 0.999), d.max())
        print("ten random re-cuts (%d distinct cut sets, thinnest slab %d planes): median %.3g p99 %.3g p99.9 %.3g max %.3g" % (
            (len(seen), min(b - a for c in seen for a, b in zip(c, c[1:]))) + q))
        assert len(seen) >= 8
        for a, b in zip(q, (4e-4, 6e-3, 5e-2, 0.3)):       # (ten free-running steps of 60-iteration solves: twice the later-step envelope of the three-step tests)
            assert a <= b, q
    finally:
        single.close()
        group.close()


def test_a_checkpoint_taken_before_a_recut_is_not_restored():
    """Checkpoint generations belong to the cut planes they were taken under: a re-cut drops them and takes a fresh one, so a restore after a re-cut goes
    back to the re-cut state (same step), never to particles and pressure planes laid out for the old ranges."""
    import blub_amd
    from tests.test_gpu_parity import _match_particles
    dim = (32, 32, 48)
    pos, vel = _blob(dim, (6, 8, 6), (26, 20, 42))
    cfg = dict(error_tolerance=0.0, max_num_iterations=60, error_check_frequency=8)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=4, binning="off", cuts=(0, 12, 24, 36, 48), movable_cuts=True)
    try:
        for f in (single, group):
            f.set_gravity_grid((0.0, -981.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        group.set_checkpoint_interval(2)
        for _ in range(3):
            single.step(util.DT)
            group.step(util.DT)
        assert group.checkpoints() == [0, 2]
        group.recut((0, 8, 28, 40, 48))
        assert group.checkpoints() == [3]                       # the generations of the old ranges are gone, one of the re-cut state (step 3) exists
        single.step(util.DT)
        group.step(util.DT)                                      # step 3 (no generation: 3 % 2 != 0)
        group.synchronize()
        group.restore(3, group.exchange_sequence() + 1024)      # back to the re-cut state ...
        group.step(util.DT)                                      # ... and step 3 again
        assert group.num_particles() == pos.shape[0] and group.cuts() == [0, 8, 28, 40, 48]
        d = _match_particles(group.get_particles()[0][:, :3].astype(np.float64), single.get_particles()[0][:, :3].astype(np.float64))
        q = (np.median(d), np.quantile(d, 0.99), np.quantile(d, 0.999), d.max())
        print("restore after a re-cut, step 4: median %.3g p99 %.3g p99.9 %.3g max %.3g" % q)
        for a, b in zip(q, (2e-4, 3e-3, 3e-2, 0.1)):
            assert a <= b, q
        with pytest.raises(blub_amd.BlubError):
            group.restore(2, group.exchange_sequence() + 1024)   # the generation of step 2 belonged to the old cuts
    finally:
        single.close()
        group.close()


def test_rebalancing_follows_the_fluid():
    """blub_slab_group_rebalance: corner_dams_128 (the metric's scene family at 128^3) as 8 slabs with UNIFORM cuts -- six slabs start without fluid --, a
    re-balance every 8 steps for 64 steps (the dams collapse and spread over z).  The cuts move, every slab ends up with fluid, the heaviest slab's share of
    the FLUID bricks stays below twice the mean, no particle is lost, and the body of water stays the single domain's (centre of mass, occupancy)."""
    import blub_amd
    from blub_amd import slab_scene
    cfg = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", "corner_dams_128.json")).config
    dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, 1)
    pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
    single = blub_amd.HybridFluid(dim, len(pos) + 64)
    group = blub_amd.SlabGroup(dim, len(pos) + 64, local=8, movable_cuts=True)
    try:
        for f in (single, group):
            f.set_gravity_grid(gravity)
            f.set_particles(pos)
        cuts0 = group.cuts()
        history, moves = [cuts0], 0
        for step in range(64):
            if step % 8 == 0:
                if group.rebalance(min_layers=1):
                    moves += 1
                    history.append(group.cuts())
            single.step(util.DT)
            group.step(util.DT)
        assert group.num_particles() == len(pos)
        bricks = [group.local_fluid(i).brick_counts()["fluid"] for i in range(8)]
        print("rebalance: %d moves, cuts %s -> %s, FLUID bricks per slab at the end %s" % (moves, cuts0, group.cuts(), bricks))
        print("rebalance: the cuts on the way: %s" % history)
        assert moves >= 2 and any(h != cuts0 for h in history)      # (the final cuts may well be the uniform ones again: the water ends up spread over all of z)
        assert history[1][1] < cuts0[1] and history[1][-2] > cuts0[-2]      # the first move pulls the outer cuts towards the two dams
        assert min(bricks) > 0 and max(bricks) <= 2.0 * sum(bricks) / 8.0, bricks
        ps = single.get_particles()[0][:, :3].astype(np.float64)
        pg = group.get_particles()[0][:, :3].astype(np.float64)
        com = np.abs(ps.mean(0) - pg.mean(0)).max()
        hs = np.histogramdd(ps, bins=(16, 16, 16), range=[(0, dim[0]), (0, dim[1]), (0, dim[2])])[0]
        hg = np.histogramdd(pg, bins=(16, 16, 16), range=[(0, dim[0]), (0, dim[1]), (0, dim[2])])[0]
        l1 = np.abs(hs - hg).sum() / len(pos)
        print("rebalance: centre of mass apart by %.3g cells, occupancy histogram L1 %.3g" % (com, l1))
        assert com < 0.3 and l1 < 0.25      # (64 free-running steps of loosely converged solves: the statistical comparison of tests/test_gpu_fullsize.py)
    finally:
        single.close()
        group.close()


def test_balanced_cuts_of_the_metric_scene_give_every_slab_fluid():
    """scenes/corner_dams_256.json (the configuration the metric is quoted on): uniform cuts into 8 leave six slabs without a FLUID brick; the balanced
    cuts give every slab an eighth.  One step of the group (direct transport) on those cuts: the PCG statistics of every slab agree with each other and
    with the single domain's iteration count, every slab owns particles, none is lost."""
    import blub_amd
    from blub_amd import slab_scene
    cfg = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", "corner_dams_256.json")).config
    dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, 1)
    pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
    uniform = [blub_amd.SlabGroup.slab_range(dim[2], 8, i)[0] for i in range(8)] + [dim[2]]
    bu = blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos, uniform)
    cuts, bb = blub_amd.SlabGroup.balanced_cuts(dim, pos, 8)
    print("corner_dams_256 as 8 slabs: uniform cuts %s -> fluid bricks %s; balanced cuts %s -> %s" % (uniform, bu, cuts, bb))
    assert sum(b == 0 for b in bu) == 6 and min(bb) > 0 and max(bb) <= 1.25 * sum(bb) / 8.0
    assert bb == blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos, cuts)
    single = blub_amd.HybridFluid(dim, len(pos) + 64)
    group = blub_amd.SlabGroup(dim, len(pos) + 64, local=8, cuts=cuts)
    try:
        assert group.transport() == "direct"
        for f in (single, group):
            f.set_gravity_grid(gravity)
            f.set_particles(pos)
        counts = [group.local_fluid(i).num_particles() for i in range(8)]
        assert min(counts) > 0 and max(counts) <= 1.3 * len(pos) / 8.0, counts
        for step in range(3):
            single.step(util.DT)
            group.step(util.DT)
            for w in (0, 1):
                st = [group.local_fluid(i).solver_stats(w) for i in range(8)]
                assert all(x == st[0] for x in st), st          # every slab derives the same scalars: the same statistics
                if step == 0:
                    # (step 0 solves the same problem up to the rounding of the gathers: the same convergence decision within one check interval; later
                    #  steps of these loosely converged solves drift apart like any two runs -- 32 against 20 iterations has been seen at step 2)
                    es, its = single.solver_stats(w)
                    assert abs(st[0][1] - its) <= 4, (st[0], (es, its))
        assert group.num_particles() == len(pos)
        ps = single.get_particles()[0][:, :3].astype(np.float64)
        pg = group.get_particles()[0][:, :3].astype(np.float64)
        from scipy.spatial import cKDTree
        d = cKDTree(ps).query(pg, k=1, workers=-1)[0]
        # diagnostic only: three free-running steps of LOOSELY converged solves are bimodal from run to run (a convergence decision that falls the other way --
        # 20 against 24 iterations -- moves the median from ~1e-4 to ~5e-3 cells: 3.5e-5 ... 7e-4 in six runs, 4.7e-3 in the seventh); the bound is below
        print("corner_dams_256, 8 balanced slabs vs single after 3 steps (loose default solves; diagnostic): median %.3g p99 %.3g max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()))
    finally:
        single.close()
        group.close()
    # The trajectory statement, with both solves CONVERGED (as tests/golden/ref_freerun does: nothing amplifies the order of the list atomics, so the bimodality
    # is gone and the bound can catch a regression of the cut / exchange path): the same three steps, particle by particle.
    single = blub_amd.HybridFluid(dim, len(pos) + 64, binning="off")
    group = blub_amd.SlabGroup(dim, len(pos) + 64, local=8, cuts=cuts, binning="off")
    try:
        for f in (single, group):
            f.set_gravity_grid(gravity)
            f.set_particles(pos)
            for w in (0, 1):
                f.set_solver_config(w, error_tolerance=1e-4, max_num_iterations=400, error_check_frequency=8)
        for step in range(3):
            single.step(util.DT)
            group.step(util.DT)
        assert group.num_particles() == len(pos)
        ps = single.get_particles()[0][:, :3].astype(np.float64)
        pg = group.get_particles()[0][:, :3].astype(np.float64)
        d = cKDTree(ps).query(pg, k=1, workers=-1)[0]
        print("corner_dams_256, 8 balanced slabs vs single after 3 steps (converged solves): median %.3g p99 %.3g max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()))
        assert np.median(d) <= 2e-4 and np.quantile(d, 0.99) <= 5e-3
    finally:
        single.close()
        group.close()
