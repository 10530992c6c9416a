"""1-rank RCCL smoke test of the slab group (create_rccl, ncclAllReduce on the engine's stream) against the plain engine."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import blub_amd
from scipy.spatial import cKDTree
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
DT = blub_amd.default_simulation_delta()
dim = (32, 32, 48)
rng = np.random.default_rng(4)
cells = np.stack(np.meshgrid(np.arange(6, 26), np.arange(8, 20), np.arange(6, 42), indexing="ij"), -1).reshape(-1, 3)
pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
g = blub_amd.SlabGroup.from_torch_distributed(dim, len(pos), device=0, binning="off")
s = blub_amd.HybridFluid(dim, len(pos), binning="off")
for f in (g, s):
    f.set_gravity_grid((0, -981.0, 0)); f.set_particles(pos)
    for w in (0, 1): f.set_solver_config(w, error_tolerance=0.0, max_num_iterations=60, error_check_frequency=8)
for _ in range(2):
    g.step(DT); s.step(DT)
d, _ = cKDTree(s.get_particles()[0][:, :3]).query(g.get_particles()[0][:, :3])
print("rccl 1-rank group vs engine: median %.3g max %.3g" % (np.median(d), d.max()))
assert np.median(d) < 2e-4 and d.max() < 0.1
g.close(); s.close(); dist.destroy_process_group(); print("rccl smoke ok")
