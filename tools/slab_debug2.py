import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
from scipy.spatial import cKDTree
DT = blub_amd.default_simulation_delta()
dim = (32, 32, 48)
rng = np.random.default_rng(4)
cells = np.stack(np.meshgrid(np.arange(6, 26), np.arange(8, 20), np.arange(6, 42), indexing="ij"), -1).reshape(-1, 3)
pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
vel = [np.zeros((pos.shape[0], 4), np.float32) for _ in range(3)]
vel[2][:, 3] = 6.0 * np.sin(pos[:, 0] * 0.4)
cfg = dict(error_tolerance=2e-6, max_num_iterations=400, error_check_frequency=8)
def mk(kind):
    if kind == "group1":
        f = blub_amd.SlabGroup(dim, pos.shape[0], local=1, binning="off")
    else:
        f = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
        if kind != "auto": f.set_pcg_work_mapping(kind)
    f.set_gravity_grid((0.0, -981.0, 0.0)); f.set_particles(pos, *vel)
    for w in (0, 1): f.set_solver_config(w, **cfg)
    return f
def run(a, b, steps=3):
    A, B = mk(a), mk(b)
    for step in range(steps):
        A.step(DT); B.step(DT)
        pa, pb = A.get_particles()[0][:, :3].astype(np.float64), B.get_particles()[0][:, :3].astype(np.float64)
        if a.startswith("group") or b.startswith("group"):
            d, _ = cKDTree(pb).query(pa, k=1)
        else:
            d = np.abs(pa - pb).max(axis=1)
        print("%s vs %s step %d: median %.3g p99 %.3g p99.9 %.3g max %.3g n>5e-4 %d" % (a, b, step, np.median(d), np.quantile(d, .99), np.quantile(d, .999), d.max(), (d > 5e-4).sum()))
    A.close(); B.close()
run("auto", "auto"); run("auto", "bricks"); run("bricks", "bricks"); run("bricks", "group1")
