"""60 steps of corner_dams_128 on the GPU and on the CPU oracle: statistical tracking (centre of mass, occupancy, kinetic energy, solver statistics)."""
import os, sys, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
from oracle.oracle import Oracle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = blub_amd.default_simulation_delta()
scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_128.json"))
f = scene.fluid()
dim = f.grid_dimension()
o = Oracle(dim[0], dim[1], dim[2], f.num_particles() + 64)
o.set_particles(f.get_particles()[0])
o.set_gravity_grid(np.float32(list(scene.config.gravity)) / np.float32(scene.config.grid_to_world_scale))
occ = lambda p: np.bincount(((p[:, 2].astype(int) * dim[1] + p[:, 1].astype(int)) * dim[0] + p[:, 0].astype(int)), minlength=int(np.prod(dim)))
for step in range(1, 61):
    scene.step(DT); o.step(DT)
    if step % 10 == 0 or step <= 3:
        f.synchronize(); f.update_statistics()
        pg = f.get_particles(); po = o.get_particles()
        a, b = pg[0][:, :3].astype(np.float64), po[0][:, :3].astype(np.float64)
        vg = np.stack([pg[1][:, 3], pg[2][:, 3], pg[3][:, 3]], 1).astype(np.float64); vo = np.stack([po[1][:, 3], po[2][:, 3], po[3][:, 3]], 1).astype(np.float64)
        com = np.abs(a.mean(0) - b.mean(0)).max(); l1 = np.abs(occ(a) - occ(b)).sum() / len(a)
        ke = (vg ** 2).sum() / (vo ** 2).sum()
        sv = f.pressure_solver_stats_velocity()[-1]; sd = f.pressure_solver_stats_density()[-1]
        print("step %2d com %.3g occL1 %.3g KE ratio %.4f | gpu it %d/%d err %.3g/%.3g | oracle it %d/%d err %.3g/%.3g" % (step, com, l1, ke, sv.iteration_count, sd.iteration_count, sv.error, sd.error, o.solver_stats(0)[1], o.solver_stats(1)[1], o.solver_stats(0)[0], o.solver_stats(1)[0]))
