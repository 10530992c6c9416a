"""Generates tests/golden/*.npz with the CPU oracle (oracle/libbluboracle.so).

The reference ships no golden vectors and cannot run here (parity unpinned), so these fixtures are
ORACLE-GENERATED: they pin today's oracle + HIP behaviour against accidental drift, they are not reference outputs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DT = float(np.float32(8333333) / np.float32(1e9))


def step_fixture():
    dim = (32, 24, 24)   # 18 432 cells (> 16 384, pressure_solver.rs:551)
    rng = np.random.default_rng(2024)
    cells = np.stack(np.meshgrid(np.arange(1, 11), np.arange(1, 9), np.arange(6, 14), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    vel = []
    for c in range(3):
        rows = np.zeros((len(pos), 4), np.float32)
        rows[:, :3] = (rng.standard_normal((len(pos), 3)) * 0.1).astype(np.float32)
        rows[:, 3] = (2.0 * np.cos(pos[:, (c + 1) % 3] * 0.5)).astype(np.float32)
        vel.append(rows)
    o = Oracle(*dim, len(pos))
    o.set_quirks(binning="off")
    o.set_gravity_grid((0.0, -981.0, 0.0))
    for w in (0, 1):
        o.set_solver_config(w, error_tolerance=0.0, max_num_iterations=160, error_check_frequency=8)   # fixed, far past convergence
    o.set_particles(pos, *vel)
    o.step(DT)
    out = o.get_particles()
    np.savez_compressed(os.path.join(HERE, "step_32x24x24.npz"), dim=np.array(dim), dt=np.float32(DT), pos_in=pos, vx_in=vel[0], vy_in=vel[1], vz_in=vel[2],
                        pos_out=out[0][:, :3], vx_out=out[1], vy_out=out[2], vz_out=out[3], marker_out=o.read_volume("marker"),
                        pressure_velocity=o.read_volume("pressure_velocity"), pressure_density=o.read_volume("pressure_density"),
                        stats=np.array([o.solver_stats(0), o.solver_stats(1)], np.float64))


def pcg_fixture():
    n = (32, 24, 24)
    rng = np.random.default_rng(7)
    marker = -np.ones(n[::-1], np.int8)
    marker[[0, -1], :, :] = 0; marker[:, [0, -1], :] = 0; marker[:, :, [0, -1]] = 0
    blob = rng.random(n[::-1]) < 0.6
    blob[:, 14:, :] = False
    marker[(marker == -1) & blob] = 1
    marker[8:12, 2:5, 10:14] = 0
    b = np.where(marker == 1, rng.standard_normal(n[::-1]), 0).astype(np.float32)
    res = {}
    for k in (0, 3, 8):
        o = Oracle(*n, 8)
        o.write_volume("marker", marker)
        o.write_volume("residual", b)
        o.set_solver_config(0, error_tolerance=0.0, max_num_iterations=k, error_check_frequency=4)
        o.run_stage("solve_velocity", DT)
        res["p_%d" % k] = o.read_volume("pressure_velocity")
        res["r_%d" % k] = o.read_volume("residual")
        res["stats_%d" % k] = np.array(o.solver_stats(0), np.float64)
    np.savez_compressed(os.path.join(HERE, "pcg_32x24x24.npz"), dim=np.array(n), dt=np.float32(DT), marker=marker, b=b, **res)


def seeding_fixture():
    o = Oracle(128, 64, 64, 1238328)
    s = np.float32(0.01)
    o.add_fluid_cube(np.float32([0, 0, 0]) / s, np.float32([0.64, 0.4, 0.64]) / s)
    p = o.get_particles()[0]
    np.savez_compressed(os.path.join(HERE, "seeding_dam_halfhalf.npz"), count=np.array(o.num_particles), first=p[:64], last=p[-64:],
                        checksum=np.array([p[:, :3].astype(np.float64).sum(), np.bitwise_xor.reduce(p[:, :3].view(np.uint32).reshape(-1))], np.float64))


def voxelize_fixture():
    """Solid voxels of two overlapping meshes (a rotated translating cube, a rotating octahedron) on a non-cubic grid."""
    from oracle.oracle import pack_mesh_desc
    dim = (40, 24, 32)
    cube = np.array([[x, y, z] for z in (-.5, .5) for y in (-.5, .5) for x in (-.5, .5)], np.float32)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (1, 3, 7, 5), (3, 2, 6, 7), (2, 0, 4, 6)]
    cube_i = np.array([[q[0], q[1], q[2], q[0], q[2], q[3]] for q in quads], np.uint32).reshape(-1)
    octa = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    octa_i = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.uint32).reshape(-1) + 8
    positions = np.concatenate([cube, octa])
    indices = np.concatenate([cube_i, octa_i])
    c, s_ = np.cos(0.6), np.sin(0.6)
    rot = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]) @ np.array([[1, 0, 0], [0, np.cos(0.3), -np.sin(0.3)], [0, np.sin(0.3), np.cos(0.3)]])
    m0 = np.concatenate([rot * 11.5, np.array([[15.2], [11.7], [14.9]])], 1)
    m1 = np.concatenate([np.eye(3) * 9.25, np.array([[26.4], [12.3], [17.6]])], 1)
    descs = np.stack([pack_mesh_desc(m0, velocity=(7.5, -2.25, 1.0), index_begin=0, index_end=36),
                      pack_mesh_desc(m1, velocity=(0.5, 0.25, -3.0), rotation_axis_scaled=(0.0, 2.5, 0.75), index_begin=36, index_end=60)])
    o = Oracle(*dim, 8)
    o.voxelize(positions, indices, descs)
    vox = o.read_volume("solid")
    zz, yy, xx = np.nonzero(vox[..., 3])
    np.savez_compressed(os.path.join(HERE, "voxelize_40x24x32.npz"), dim=np.array(dim), positions=positions, indices=indices, descs=descs,
                        solid_zyx=np.stack([zz, yy, xx], 1).astype(np.int16), solid_velocity=vox[zz, yy, xx, :3])


if __name__ == "__main__":
    voxelize_fixture()
    step_fixture()
    pcg_fixture()
    seeding_fixture()
    print("golden fixtures written to", HERE)
