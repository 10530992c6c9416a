"""Static objects on the GPU: the HIP conservative-hull voxeliser against the oracle's restatement of
scene/voxelization.rs + shader/voxelize/conservative_hull.{vert,frag}, and a scene with a moving solid end to end."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import util
from tests.conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

DELTA_NS = 8333333


def _icosphere(subdiv=2):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.uint32).reshape(-1)


def _desc(m3x4, velocity=(0, 0, 0), axis=(0, 0, 0), begin=0, end=0):
    import blub_amd
    d = blub_amd.MeshDesc()
    for r in range(3):
        for c in range(4):
            d.voxel_transform[r][c] = float(m3x4[r][c])
    for k in range(3):
        d.fluid_space_velocity[k] = float(velocity[k])
        d.fluid_space_rotation_axis_scaled[k] = float(axis[k])
    d.index_begin, d.index_end = int(begin), int(end)
    return d


def _pack(d):
    return orc.pack_mesh_desc(np.array([list(r) for r in d.voxel_transform], np.float32), list(d.fluid_space_velocity), list(d.fluid_space_rotation_axis_scaled), d.index_begin, d.index_end)


def _rot(axis, deg):
    from scipy.spatial.transform import Rotation as R
    a = np.asarray(axis, np.float64)
    return R.from_rotvec(a / np.linalg.norm(a) * np.deg2rad(deg)).as_matrix()


@pytest.mark.parametrize("dim", [(32, 32, 32), (48, 32, 24)])
def test_voxeliser_matches_oracle(dim):
    """Solid flags: bit-exact (same f32 operation order, no contraction).  Velocities: exact for a translating mesh; for a
    rotating one several fragments write one voxel with positions that differ inside the cell (conservative_hull.frag:36 vs
    :47-52) and the GPU's store order is not the oracle's, so they agree to |axis| * (cell diagonal + 1)."""
    import blub_amd
    nx, ny, nz = dim
    rng = np.random.default_rng(7)
    cube_p, cube_i = blub_amd.load_obj(os.path.join(ROOT, "scenes", "models", "unit_cube.obj"))
    sph_p, sph_i = _icosphere(3)
    soup_p = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    soup_i = np.arange(300, dtype=np.uint32)
    positions = np.concatenate([cube_p, sph_p, soup_p])
    indices = np.concatenate([cube_i, sph_i + len(cube_p), soup_i + len(cube_p) + len(sph_p)])
    r0, r1, r2 = len(cube_i), len(cube_i) + len(sph_i), len(indices)

    def xf(rot, scale, t):
        m = np.zeros((3, 4))
        m[:, :3] = rot * scale
        m[:, 3] = t
        return m
    cases = {
        "cube, rotated, translating": [_desc(xf(_rot((1, 2, 3), 37.0), 9.3, (nx * 0.4, ny * 0.5, nz * 0.5)), velocity=(12.5, -3.0, 0.25), begin=0, end=r0)],
        "sphere, rotating": [_desc(xf(np.eye(3), 7.7, (nx * 0.6, ny * 0.45, nz * 0.5)), velocity=(1, 2, 3), axis=(0.2, 1.5, -0.3), begin=r0, end=r1)],
        "soup sticking out of the grid": [_desc(xf(_rot((0, 1, 1), 20.0), max(dim) * 0.6, (nx * 0.5, ny * 0.5, nz * 0.5)), begin=r1, end=r2)],
        "all three, overlapping": [_desc(xf(_rot((1, 0, 0), 10.0), 8.0, (nx * 0.5, ny * 0.5, nz * 0.5)), velocity=(1, 0, 0), begin=0, end=r0),
                                   _desc(xf(np.eye(3), 6.0, (nx * 0.55, ny * 0.5, nz * 0.5)), velocity=(0, 2, 0), begin=r0, end=r1),
                                   _desc(xf(np.eye(3), 5.0, (nx * 0.5, ny * 0.55, nz * 0.5)), velocity=(0, 0, 3), begin=r1, end=r2)],
    }
    o = orc.Oracle(nx, ny, nz, 8)
    h = blub_amd.HybridFluid(dim, 8)
    try:
        h.set_meshes(positions, indices)
        for name, descs in cases.items():
            o.voxelize(positions, indices, [_pack(d) for d in descs])
            h.voxelize(descs)
            vo, vh = o.read_volume("solid"), h.read_volume("solid")
            so, sh = vo[..., 3] == 1.0, vh[..., 3] == 1.0
            print("%-32s %s: %d solid voxels" % (name, dim, so.sum()))
            assert so.sum() > 50
            assert np.array_equal(vh[..., 3], vo[..., 3]), (name, (so != sh).sum())
            rotating = any(any(d.fluid_space_rotation_axis_scaled) for d in descs)
            if not rotating:
                assert np.array_equal(vh, vo), name
            else:
                amax = max(np.linalg.norm(list(d.fluid_space_rotation_axis_scaled)) for d in descs)
                assert np.abs(vh[..., :3] - vo[..., :3]).max() <= amax * (3 ** 0.5 + 1.0) + 0.02, name
            # the static marker pattern follows the solid volume (transfer_set_boundary_marker.comp:11-19)
            mk = h.read_volume("marker")
            assert np.all(mk[sh] == 0)
        h.voxelize([])
        assert not h.read_volume("solid").any()
    finally:
        h.close()


def test_voxelize_argument_checks():
    import blub_amd
    from blub_amd.hybrid_fluid import BlubError
    h = blub_amd.HybridFluid((32, 32, 32), 8)
    try:
        with pytest.raises(BlubError):
            h.set_meshes(np.zeros((3, 3), np.float32), np.array([0, 1, 3], np.uint32))       # index out of range
        with pytest.raises(BlubError):
            h.set_meshes(np.zeros((3, 3), np.float32), np.array([0, 1], np.uint32))          # not a multiple of 3
        h.set_meshes(np.zeros((3, 3), np.float32), np.array([0, 1, 2], np.uint32))
        with pytest.raises(BlubError):
            h.voxelize([_desc(np.eye(3, 4), begin=0, end=6)])                                # range outside the index buffer
        h.voxelize([_desc(np.eye(3, 4), begin=0, end=3)])                                    # degenerate triangle: nothing
        assert not h.read_volume("solid").any()
    finally:
        h.close()


SCENE = {
    "gravity": {"x": 0.0, "y": -9.81, "z": 0.0},
    "fluid": {"world_position": {"x": 0.0, "y": 0.0, "z": 0.0}, "max_num_particles": 40000, "grid_to_world_scale": 0.02,
              "grid_dimension": {"x": 48, "y": 32, "z": 32},
              "fluid_cubes": [{"min": {"x": 0.0, "y": 0.0, "z": 0.0}, "max": {"x": 0.4, "y": 0.24, "z": 0.64}}]},
    "static_objects": [{"model": "unit_cube.obj", "world_position": {"x": 0.68, "y": 0.16, "z": 0.32}, "scale": 0.3,
                        "rotation_angles": {"x": 0.0, "y": 0.0, "z": 0.0},
                        "animation": {"translation": {"target": {"x": 0.3, "y": 0.16, "z": 0.32}, "curve": "Linear", "duration": 0.2}}}],
}


def test_scene_with_a_moving_solid_matches_oracle():
    """A `wavegenerator`-style scene (reference: scenes/wavegenerator.json -- its cube.obj is a git-lfs pointer without data, so
    the mesh here is scenes/models/unit_cube.obj): a cube ploughs into the fluid at 1.9 m/s.  Scene::step order
    (scene/mod.rs:166-213): animate models, voxelise, fluid step.  Both sides take converged solves, the oracle is driven with
    the same mesh descriptors."""
    import blub_amd
    scene = blub_amd.Scene(text=json.dumps(SCENE))
    scene.models_dir = os.path.join(ROOT, "scenes", "models")
    f = scene.fluid()
    nx, ny, nz = f.grid_dimension()
    o = orc.Oracle(nx, ny, nz, 40000)
    try:
        pos0 = f.get_particles()[0]
        o.set_particles(pos0)
        o.set_gravity_grid((0.0, -9.81 / 0.02, 0.0))
        for w in (0, 1):
            f.set_solver_config(w, error_tolerance=2e-6, max_num_iterations=600, error_check_frequency=8)
            o.set_solver_config(w, 2e-6, 600, 8)
        f.particle_rebinning_step_frequency = 0
        o.set_rebinning_frequency(0)
        mesh_p, mesh_i = blub_amd.load_obj(os.path.join(scene.models_dir, "unit_cube.obj"))
        total = 0
        for step in range(1, 7):
            scene.step(util.DT)
            total += DELTA_NS
            d = blub_amd.mesh_desc_at_time(scene.config, 0, total, DELTA_NS)
            d.index_begin, d.index_end = 0, len(mesh_i)
            o.voxelize(mesh_p, mesh_i, [_pack(d)])
            o.step(util.DT)
            assert scene.total_simulated_time_ns == total
            so, sh = o.read_volume("solid"), f.read_volume("solid")
            assert np.array_equal(sh, so), step
            if step == 1:
                assert not so[..., :3].any()                       # first step: no velocity yet (models.rs:195)
            else:
                assert np.allclose(so[so[..., 3] == 1.0][:, 0], np.float16(-1.9 / 0.02), rtol=2e-3)
            po, ph = o.get_particles()[0], f.get_particles()[0]
            dd = np.abs(ph[:, :3] - po[:, :3]).max(axis=1)
            print("step %d: %d solid voxels; particles vs oracle: median %.3g p99 %.3g max %.3g" % (step, (so[..., 3] == 1).sum(), np.median(dd), np.quantile(dd, 0.99), dd.max()))
            assert np.median(dd) < 1e-4 and np.quantile(dd, 0.99) < 5e-3 and dd.max() < 0.5, step
        mo, mh = o.read_volume("marker"), f.read_volume("marker")
        assert (mo != mh).mean() < 1e-3
        # the solid did push fluid: particles in front of the cube move in -x faster than anything a dam break alone produces
        vx = f.get_particles()[1][:, 3]
        front = (ph[:, 0] > 20) & (ph[:, 0] < 27)
        assert front.sum() > 100 and vx[front].min() < -40.0, (front.sum(), vx[front].min() if front.any() else None)
        # no particle ends inside a solid voxel (advect_particles.comp:46-65, 134-173)
        cells = np.floor(ph[:, :3]).astype(int)
        inside = sh[cells[:, 2], cells[:, 1], cells[:, 0], 3] == 1.0
        assert inside.mean() < 0.01, inside.mean()
    finally:
        f.close()


@pytest.mark.parametrize("recut", [False, True])
def test_moving_solid_in_a_z_slab_group_matches_single_domain(recut):
    """The voxelised solid straddles the interface of two z-slabs (every slab voxelises the meshes in global coordinates);
    the group must stay inside the engine's run-to-run noise envelope of the single-domain run (see
    test_gpu_parity.py::test_z_slab_decomposition_matches_single_domain for the envelope).  recut: slabs that hold the whole grid, the cut plane moves from
    z = 16 to z = 12 after the second step -- through the moving solid: the solid volume is every slab's own (voxelised in global coordinates), only the
    pressure planes and the particles change owner (blub_slab_group_recut)."""
    import blub_amd
    from scipy.spatial import cKDTree
    scene = blub_amd.Scene(text=json.dumps(SCENE))
    scene.models_dir = os.path.join(ROOT, "scenes", "models")
    single = scene.fluid()
    dim = single.grid_dimension()
    pos0 = single.get_particles()[0]
    group = blub_amd.SlabGroup(dim, 40000, local=2, movable_cuts=recut)
    try:
        mesh_p, mesh_i = blub_amd.load_obj(os.path.join(scene.models_dir, "unit_cube.obj"))
        group.set_meshes(mesh_p, mesh_i)
        group.set_gravity_grid((0.0, -9.81 / 0.02, 0.0))
        group.set_particles(pos0)
        cfg = dict(error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)   # no convergence decision: see the slab test
        for w in (0, 1):
            single.set_solver_config(w, **cfg)
            group.set_solver_config(w, **cfg)
        single.particle_rebinning_step_frequency = 0
        group.set_rebinning_frequency(0)
        total = 0
        for step in range(1, 5):
            scene.step(util.DT)
            total += DELTA_NS
            d = blub_amd.mesh_desc_at_time(scene.config, 0, total, DELTA_NS)
            d.index_begin, d.index_end = 0, len(mesh_i)
            if recut and step == 3:
                group.recut((0, 12, 32))
                assert group.cuts() == [0, 12, 32]
            group.voxelize([d])
            group.step(util.DT)
            ps = single.get_particles()[0][:, :3].astype(np.float64)
            pg = group.get_particles()[0][:, :3].astype(np.float64)
            assert pg.shape == ps.shape
            dd, idx = cKDTree(ps).query(pg, k=1)
            assert len(np.unique(idx)) == len(pg)
            print("step %d: slabs vs single: median %.3g p99 %.3g max %.3g" % (step, np.median(dd), np.quantile(dd, 0.99), dd.max()))
            assert np.median(dd) < 2e-4 and np.quantile(dd, 0.99) < 3e-3 and dd.max() < 0.1
        for i in range(2):
            assert np.array_equal(group.local_fluid(i).read_volume("solid"), single.read_volume("solid"))
    finally:
        single.close()
        group.close()
