"""Development probe: a long run of a scene -- finite particles inside the domain, solver statistics sane, no tail time-outs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, steps = (sys.argv[1] if len(sys.argv) > 1 else "corner_dams_256"), int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dt = blub_amd.default_simulation_delta()
scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", name + ".json"))
f = scene.fluid()
t0 = time.perf_counter()
for s in range(steps):
    scene.step(dt)
    if s % 64 == 63:
        f.update_statistics()
f.synchronize()
el = time.perf_counter() - t0
p = f.get_particles()[0][:, :3]
nx, ny, nz = f.grid_dimension()
sv, sd = f.pressure_solver_stats_velocity(), f.pressure_solver_stats_density()
print("%s: %d steps in %.2f s (%.0f steps/s); finite %s; inside %s; y range %.2f..%.2f; last velocity stats %s density %s; bricks %s" % (
    name, steps, el, steps / el, bool(np.all(np.isfinite(p))), bool((p.min() >= 1.0) and (p[:, 0].max() <= nx - 1) and (p[:, 1].max() <= ny - 1) and (p[:, 2].max() <= nz - 1)),
    p[:, 1].min(), p[:, 1].max(), [(round(s.error, 4), s.iteration_count) for s in sv[-3:]], [(round(s.error, 4), s.iteration_count) for s in sd[-3:]], f.brick_counts()))
assert all(s.iteration_count >= 0 for s in sv) and all(s.iteration_count >= 0 for s in sd)
