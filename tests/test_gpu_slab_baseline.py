"""The z-slab group on the BASELINE multi-GPU configurations themselves (round-3 review, missing item 1), all slabs on the one GPU of the
test box (loopback transport):

  * BASELINE.json configs[3]: scenes/dam_halfhalf_highres.json (256x128x128, 10 113 264 particles) as 2 and as 4 z-slabs,
  * BASELINE.json configs[4]: scenes/corner_dams_512.json (512^3, 8 065 008 particles) as 8 z-slabs,

against the single-domain engine on the same particles.  Step 0 segment by segment (blub_slab_group_run_stages): the marker of every
slab's own planes EXACTLY; the P2G velocities on every own plane -- interface planes included -- within 1e-5 * max(1, |v|) (the gathers
add the same particles in another order); a PCG solve with a fixed iteration count within 3e-4 of the pressure scale with identical
statistics on every slab and the single domain's iteration count; the migration conserves the multiset of particle records and leaves
every slab with particles of its own z-range only.  Then three free-running steps inside the run-to-run noise envelope of
tests/test_gpu_parity.py::test_z_slab_decomposition_matches_single_domain.

Also here: a fluid front that reaches an interface which carried nothing before (round-3 ADVICE: the host-synchronisation-free
exchange sized its messages from the previous step and LOST what did not fit), and switching the asynchronous exchange off mid-run.
"""
import os
import time

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _records(p):
    """the multiset of particle records (position + three APIC rows; the list pointer is scratch) as a sorted array of 64-bit hashes"""
    a = np.concatenate([np.ascontiguousarray(p[0][:, :3]).view(np.uint32)] + [np.ascontiguousarray(x).view(np.uint32) for x in p[1:]], axis=1).astype(np.uint64)
    mult = np.random.default_rng(99).integers(1, 2 ** 63, size=a.shape[1], dtype=np.uint64) | np.uint64(1)
    h = np.zeros(len(a), np.uint64)
    for k in range(a.shape[1]):
        h = (h ^ (a[:, k] * mult[k])) * np.uint64(0x9E3779B97F4A7C15)
    return np.sort(h)


def _match_sample(pg, ps, n=300000, seed=3):
    """nearest single-domain particle of a random sample of the group's particles (the full one-to-one matching of 10 M points takes minutes)"""
    from scipy.spatial import cKDTree
    idx = np.random.default_rng(seed).choice(len(pg), size=min(n, len(pg)), replace=False)
    d, j = cKDTree(ps).query(pg[idx], k=1, workers=-1)
    assert len(np.unique(j)) == len(idx), "matching is not one-to-one"
    return d


def _own(group, i, name):
    z0, z1 = group.local_range(i)
    return group.local_fluid(i).read_volume(name)[z0:z1], (z0, z1)


def _quantiles(a):
    return (np.median(a), np.quantile(a, 0.99), np.quantile(a, 0.999), a.max())


@pytest.mark.parametrize("scene_name,slabs,particles", [("dam_halfhalf_highres", 2, 10113264), ("dam_halfhalf_highres", 4, 10113264), ("corner_dams_512", 8, 8065008)])
def test_slab_group_on_the_baseline_multi_gpu_configurations(scene_name, slabs, particles):
    import blub_amd
    t0 = time.time()
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", scene_name + ".json"))
    single = scene.fluid()
    dim = single.grid_dimension()
    p_in = single.get_particles()
    assert single.num_particles() == particles
    gravity = np.float32(list(scene.config.gravity)) / np.float32(scene.config.grid_to_world_scale)
    single.particle_rebinning_step_frequency = 0
    K = 8
    group = blub_amd.SlabGroup(dim, particles + 64, local=slabs, binning="off")
    try:
        group.set_gravity_grid(gravity)
        group.set_rebinning_frequency(0)
        group.set_particles(p_in[0])
        for f in (single, group):
            for w in (0, 1):
                f.set_solver_config(w, error_tolerance=0.0, max_num_iterations=K, error_check_frequency=4)
        ranges = [group.local_range(i) for i in range(slabs)]
        assert ranges[0][0] == 0 and ranges[-1][1] == dim[2] and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        counts0 = [group.local_fluid(i).num_particles() for i in range(slabs)]
        assert sum(counts0) == particles
        # ---- P2G
        single.run_stage("transfer", util.DT)
        group.run_stages(util.DT, "ghosts", "transfer")
        m_single = single.read_volume("marker")
        fluid = m_single == 1
        for i in range(slabs):
            m, (z0, z1) = _own(group, i, "marker")
            assert np.array_equal(m, m_single[z0:z1]), "slab %d: %d marker cells differ" % (i, (m != m_single[z0:z1]).sum())
        for v in ("vel_x", "vel_y", "vel_z"):
            ref = single.read_volume(v)
            for i in range(slabs):
                a, (z0, z1) = _own(group, i, v)
                util.assert_close("%s on the planes [%d, %d) of slab %d" % (v, z0, z1, i), a, ref[z0:z1], rel=1e-5)
                for z in (z0, z1 - 1):   # the interface planes carry fluid in these scenes: the comparison above is not vacuous there
                    if v == "vel_y" and 0 < z < dim[2] - 1 and fluid[z].any():
                        assert np.abs(ref[z]).max() > 0
        # ---- divergence + a PCG solve with K fixed iterations
        single.run_stage("divergence", util.DT)
        single.run_stage("solve_velocity", util.DT)
        group.run_stages(util.DT, "divergence", "solve_velocity")
        p_ref = single.read_volume("pressure_velocity")
        scale = np.abs(p_ref[fluid]).max()
        assert scale > 0
        for i in range(slabs):
            a, (z0, z1) = _own(group, i, "pressure_velocity")
            util.assert_close("pressure on slab %d" % i, a[fluid[z0:z1]], p_ref[z0:z1][fluid[z0:z1]], abs_=3e-4 * scale)
        e_s, it_s = single.solver_stats(0)
        stats = [group.local_fluid(i).solver_stats(0) for i in range(slabs)]
        assert all(s == stats[0] for s in stats), stats                                   # identical statistics on every slab
        assert stats[0][1] == it_s == K and abs(stats[0][0] - e_s) <= 1e-3 * e_s, (stats[0], (e_s, it_s))
        # ---- projection, advection, then the migration by itself
        single.run_stage("project", util.DT)
        single.run_stage("advect", util.DT)
        group.run_stages(util.DT, "binning", "advect")
        before = _records(group.get_particles())
        group.run_stages(util.DT, "migrate", "migrate")
        pg = group.get_particles()
        after = _records(pg)
        assert before.shape == after.shape == (particles,) and np.array_equal(before, after), "the migration changed the multiset of particle records"
        counts1 = [group.local_fluid(i).num_particles() for i in range(slabs)]
        assert sum(counts1) == particles
        off = 0
        for i, (z0, z1) in enumerate(ranges):
            z = pg[0][off:off + counts1[i], 2]
            off += counts1[i]
            assert np.all(z >= z0) and np.all(z < z1), "slab %d holds particles outside [%d, %d)" % (i, z0, z1)
        moved = sum(abs(a - b) for a, b in zip(counts0, counts1))
        # ---- the rest of step 0, then free-running steps against the single domain and its own rerun noise
        for st in ("density_gather", "solve_density", "position_change", "correct"):
            single.run_stage(st, util.DT)
        single.step_counter = 1
        group.run_stages(util.DT, "density_gather", "finish")
        rerun = blub_amd.HybridFluid(dim, particles + 64, binning="off")
        try:
            rerun.set_gravity_grid(gravity)
            rerun.set_particles(p_in[0])
            for w in (0, 1):
                rerun.set_solver_config(w, error_tolerance=0.0, max_num_iterations=K, error_check_frequency=4)
            rerun.step(util.DT)
            for step in range(3):
                if step:
                    for f in (single, rerun, group):
                        # (one engine at a time: the brick-list build is ONE co-resident launch -- 256 workgroups of 1024 threads at 512^3 --, and three
                        #  of them from three streams do not fit the device together; they would sit out each other's spin bound)
                        f.step(util.DT)
                        f.synchronize()
                if step == 1:
                    continue
                ps = single.get_particles()[0][:, :3].astype(np.float64)
                pr = rerun.get_particles()[0][:, :3].astype(np.float64)
                pgs = group.get_particles()[0][:, :3].astype(np.float64)
                assert pgs.shape == ps.shape
                qf = _quantiles(np.abs(pr - ps).max(axis=1))
                qd = _quantiles(_match_sample(pgs, ps))
                print("%s / %d slabs, step %d: slabs vs single median %.3g p99 %.3g p99.9 %.3g max %.3g | rerun floor %.3g %.3g %.3g %.3g" % ((scene_name, slabs, step) + qd + qf))
                bounds = (3e-5, 4e-4, 1.5e-3, 3e-3) if step == 0 else (2e-4, 3e-3, 3e-2, 0.1)
                for k in range(4):    # the envelope of the small-grid test, or twice this scene's own rerun noise where that is larger
                    assert qd[k] <= max(bounds[k], 2.0 * qf[k]), (step, qd, qf, bounds)
        finally:
            rerun.close()
        assert group.held_back() == 0
        for w in (0, 1):
            st = [group.local_fluid(i).solver_stats(w) for i in range(slabs)]
            assert all(x == st[0] for x in st)
        print("%s as %d slabs: %d particles changed slab in step 0's first migration; %.1f s" % (scene_name, slabs, moved, time.time() - t0))
    finally:
        single.close()
        group.close()


def _front_scene():
    """A block that lies entirely inside slab 0 of 2 (z < 24 of 48) and moves towards the interface at ~0.9 cells per step: the interface
    carries NOTHING for the first steps, then a front of ~2 000 particles per step and plane arrives at once."""
    dim = (48, 32, 48)
    rng = np.random.default_rng(11)
    cells = np.stack(np.meshgrid(np.arange(4, 44), np.arange(2, 22), np.arange(8, 20), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    vel = [np.zeros((pos.shape[0], 4), np.float32) for _ in range(3)]
    vel[2][:, 3] = 110.0          # cells / s: 0.92 cells per step towards +z
    return dim, pos, vel


def test_a_front_arriving_at_an_empty_interface_loses_no_particle():
    """Round-3 ADVICE (high): messages of the host-synchronisation-free exchange are sized 1.5 x the previous step's count + 2048; a link
    that carried nothing can take 2048 particles, the front brings ~7 000 per step.  What does not fit is now HELD BACK at the sender
    for one exchange (blub_slab_group_held_back counts it) instead of being dropped: every particle is conserved, nobody sits in the wrong
    slab afterwards, the group never returns an error, and it stays close to the single domain."""
    import blub_amd
    dim, pos, vel = _front_scene()
    cfg = dict(error_tolerance=0.0, max_num_iterations=60, error_check_frequency=8)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=2, binning="off")
    group.set_transport("host")       # (messages with a size only exist with host-issued transport operations; the direct transport: next test)
    try:
        for f in (single, group):
            f.set_gravity_grid((0.0, 0.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        assert group.local_fluid(1).num_particles() == 0
        arrived = []
        for step in range(14):
            single.step(util.DT)
            group.step(util.DT)
            if step in (2, 5, 8, 11, 13):
                assert group.num_particles() == pos.shape[0], "step %d: particles lost" % step      # (also brings the counts to the host)
                arrived.append(group.local_fluid(1).num_particles())
        held = group.held_back()
        print("front: particles in the upper slab after steps 2, 5, 8, 11, 13: %s; held back / left out for one exchange: %d" % (arrived, held))
        assert arrived[0] == 0 and arrived[-1] > 20000, arrived          # nothing for the first steps, then the front
        assert held > 0, "the scene did not outgrow a message: it does not test the hold-back path"
        assert group.host_syncs()[0] == 4                                 # still no host synchronisation after the first step
        pg = group.get_particles()[0]
        n0 = group.local_fluid(0).num_particles()
        z0, z1 = group.local_range(0)
        assert np.all(pg[:n0, 2] < z1) and np.all(pg[n0:, 2] >= z1)
        # (a held-back particle waits at the interface for one exchange: the group is no longer the single domain particle by particle, but it
        #  stays the same body of water)
        from scipy.spatial import cKDTree
        d = cKDTree(single.get_particles()[0][:, :3].astype(np.float64)).query(pg[:, :3].astype(np.float64), k=1)[0]
        com_g, com_s = pg[:, :3].astype(np.float64).mean(0), single.get_particles()[0][:, :3].astype(np.float64).mean(0)
        print("front: slabs vs single after 14 steps: nearest-particle distance median %.3g p99 %.3g max %.3g; centre of mass apart by %.3g cells" % (
            np.median(d), np.quantile(d, 0.99), d.max(), np.abs(com_g - com_s).max()))
        assert np.median(d) < 0.2 and np.abs(com_g - com_s).max() < 0.1
    finally:
        single.close()
        group.close()


def test_a_front_over_the_direct_transport_needs_no_hold_back():
    """The same front with the direct transport: the sender writes what travels straight into staging buffers of full particle capacity, so
    there is no message to outgrow -- nothing held back, and the group stays the single domain particle by particle."""
    import blub_amd
    from tests.test_gpu_parity import _match_particles
    dim, pos, vel = _front_scene()
    cfg = dict(error_tolerance=0.0, max_num_iterations=60, error_check_frequency=8)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=2, binning="off")
    try:
        assert group.transport() == "direct"
        for f in (single, group):
            f.set_gravity_grid((0.0, 0.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        for step in range(14):
            single.step(util.DT)
            group.step(util.DT)
        assert group.num_particles() == pos.shape[0] and group.local_fluid(1).num_particles() > 20000
        assert group.held_back() == 0 and group.host_syncs() == (0, 0)
        d = _match_particles(group.get_particles()[0][:, :3].astype(np.float64), single.get_particles()[0][:, :3].astype(np.float64))
        print("front, direct transport: slabs vs single after 14 steps: median %.3g p99 %.3g max %.3g" % (np.median(d), np.quantile(d, 0.99), d.max()))
        assert np.median(d) < 1e-3 and np.quantile(d, 0.99) < 2e-2
    finally:
        single.close()
        group.close()


def test_switching_the_asynchronous_exchange_off_mid_run():
    """Round-3 ADVICE (medium): after asynchronous steps the host-side particle counts are bounds; the synchronous protocol takes them as
    exact.  blub_slab_group_set_async_exchange(g, 0) now fetches the counts first."""
    import blub_amd
    from tests.multirank_worker import scene
    from tests.test_gpu_parity import _match_particles
    dim, pos, vel, cfg = scene()
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    group = blub_amd.SlabGroup(dim, pos.shape[0], local=3, binning="off")
    group.set_transport("host")
    try:
        for f in (single, group):
            f.set_gravity_grid((0.0, -981.0, 0.0))
            f.set_particles(pos, *vel)
            for w in (0, 1):
                f.set_solver_config(w, **cfg)
        for step in range(6):
            if step == 3:
                group.set_async_exchange(False)       # (no synchronize() / num_particles() call in between)
            if step == 5:
                group.set_async_exchange(True)
            single.step(util.DT)
            group.step(util.DT)
        assert group.num_particles() == pos.shape[0]
        assert group.host_syncs()[0] == 4 + 2 * 4              # step 0 and the two synchronous steps (whose counts are the history step 5 sizes its messages from)
        d = _match_particles(group.get_particles()[0][:, :3].astype(np.float64), single.get_particles()[0][:, :3].astype(np.float64))
        assert np.median(d) < 1e-3 and np.quantile(d, 0.99) < 2e-2, (np.median(d), np.quantile(d, 0.99))
    finally:
        single.close()
        group.close()


def test_a_slab_allocates_its_own_planes_only():
    """Round-3 review, missing item 4: every slab of a group used to allocate all sixteen grid volumes at FULL grid size.  Now it holds its own
    planes plus two brick layers on either side: eight slabs of a 256^3 grid (32 own planes + 16 each) take well under twice one full set of
    volumes instead of eight times, and what a slab does not hold reads back as zero."""
    import torch
    import blub_amd
    dim = (256, 256, 256)
    n = 256 * 256 * 256

    def used(make):
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        obj = make()
        obj.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        return obj, free0 - free1

    single, b_single = used(lambda: blub_amd.HybridFluid(dim, 1024, binning="off"))
    try:
        group, b_group = used(lambda: blub_amd.SlabGroup(dim, 1024, local=8))
        try:
            volumes = 15.5 * 4 * n          # 12 float / uint volumes, marker + descriptor bytes, the three single-reduction vectors
            print("device memory: single domain %.0f MiB, eight slabs %.0f MiB (full-size volumes: %.0f MiB)" % (b_single / 2 ** 20, b_group / 2 ** 20, volumes / 2 ** 20))
            assert b_single > 0.7 * volumes
            assert b_group < 1.9 * b_single + 8 * 64 * 2 ** 20
            z0, z1 = group.local_range(3)
            f = group.local_fluid(3)
            m = f.read_volume("marker")
            assert m.shape == (256, 256, 256)
            assert np.all(m[:z0 - 8] == 0) and np.all(m[z1 + 8:] == 0)                  # not held: zero
            assert np.all(m[z0 - 8:z1 + 8, 1:-1, 1:-1] == -1)                            # held: the static pattern (AIR inside the SOLID shell)
            v = np.zeros(dim[::-1], np.float32)
            v[:] = np.arange(256, dtype=np.float32)[:, None, None]
            f.write_volume("vel_x", v)
            back = f.read_volume("vel_x")
            assert np.array_equal(back[z0 - 8:z1 + 8], v[z0 - 8:z1 + 8]) and not back[:z0 - 8].any() and not back[z1 + 8:].any()
        finally:
            group.close()
    finally:
        single.close()
