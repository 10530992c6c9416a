import sys, os, json
sys.path.insert(0, '/root/repo')
import blub_amd
dt = blub_amd.default_simulation_delta()
s = blub_amd.Scene(path='/root/repo/scenes/corner_dams_256.json'); f = s.fluid()
f.set_pcg_schedule("single_reduction")
out = {0: [], 1: []}
for i in range(131):
    s.step(dt); f.synchronize()
    out[0].append(f.pressure_solver_stats_velocity()[-1].iteration_count); out[1].append(f.pressure_solver_stats_density()[-1].iteration_count)
print(json.dumps(out))
