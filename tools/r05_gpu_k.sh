#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
python - <<P
import blub_amd, os
sc=blub_amd.Scene(path="scenes/corner_dams_256.json"); f=sc.fluid(); dt=blub_amd.default_simulation_delta()
rows=[]
for s in range(40):
    sc.step(dt); f.synchronize()
    v=f.pressure_solver_stats_velocity()[-1]; d=f.pressure_solver_stats_density()[-1]
    rows.append((v.iteration_count, d.iteration_count))
print("velocity:", [r[0] for r in rows])
print("density: ", [r[1] for r in rows])
P
