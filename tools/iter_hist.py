"""Development probe: iteration counts of the two solves over the headline window (how many launched PCG kernels find the solve finished)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt = blub_amd.default_simulation_delta()
scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", (sys.argv[1] if len(sys.argv) > 1 else "corner_dams_256") + ".json"))
f = scene.fluid()
v, d = [], []
for s in range(130):
    scene.step(dt)
    if s % 50 == 49 or s == 129:
        f.synchronize(); f.update_statistics()
        v += [x.iteration_count for x in f.pressure_solver_stats_velocity()][-50 if s != 129 else -30:]
        d += [x.iteration_count for x in f.pressure_solver_stats_density()][-50 if s != 129 else -30:]
print("velocity:", v)
print("density :", d)
print("mean velocity %.1f density %.1f; finished launches per step %.1f of 66" % (sum(v) / len(v), sum(d) / len(d), 66 - (sum(v) + sum(d)) / len(v) - 2))
