import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
from scipy.spatial import cKDTree
DT = blub_amd.default_simulation_delta()
slabs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dim = (32, 32, 48)
rng = np.random.default_rng(4)
cells = np.stack(np.meshgrid(np.arange(6, 26), np.arange(8, 20), np.arange(6, 42), indexing="ij"), -1).reshape(-1, 3)
pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
vel = [np.zeros((pos.shape[0], 4), np.float32) for _ in range(3)]
vel[2][:, 3] = 6.0 * np.sin(pos[:, 0] * 0.4)
cfg = dict(error_tolerance=2e-6, max_num_iterations=400, error_check_frequency=8)
single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
group = blub_amd.SlabGroup(dim, pos.shape[0], local=slabs, binning="off")
for f in (single, group):
    f.set_gravity_grid((0.0, -981.0, 0.0)); f.set_particles(pos, *vel)
for w in (0, 1):
    single.set_solver_config(w, **cfg); group.set_solver_config(w, **cfg)
ranges = [group.local_range(i) for i in range(slabs)]
print("ranges", ranges)
for step in range(4):
    single.step(DT); group.step(DT)
    ps, pg = single.get_particles(), group.get_particles()
    a, b = pg[0][:, :3].astype(np.float64), ps[0][:, :3].astype(np.float64)
    d, idx = cKDTree(b).query(a, k=1)
    bad = d > 5e-4
    print("step %d: median %.3g p99 %.3g p99.9 %.3g max %.3g; n>5e-4: %d; their z: %s ; stats single %s group %s" % (
        step, np.median(d), np.quantile(d, .99), np.quantile(d, .999), d.max(), bad.sum(),
        np.round(np.histogram(a[bad, 2], bins=12, range=(0, 48))[0]) if bad.any() else "-",
        single.solver_stats(0), group.local_fluid(0).solver_stats(0)))
    for name in ("vel_x", "vel_y", "vel_z", "pressure_velocity", "pressure_density", "marker"):
        vs = single.read_volume(name)
        vg = np.zeros_like(vs)
        for i, (z0, z1) in enumerate(ranges):
            vg[z0:z1] = group.local_fluid(i).read_volume(name)[z0:z1]
        diff = np.abs(vg.astype(np.float64) - vs.astype(np.float64))
        zmax = np.unravel_index(diff.argmax(), diff.shape)
        print("   %-18s max|diff| %.3g at (z,y,x)=%s  scale %.3g" % (name, diff.max(), zmax, np.abs(vs).max()))
