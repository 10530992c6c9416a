#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
python - <<P
import blub_amd, numpy as np
from blub_amd import slab_scene
dt=blub_amd.default_simulation_delta()
cfg=blub_amd.Scene.parse(path="scenes/corner_dams_256.json").config
dim0,scale,gravity,cubes,maxp=slab_scene.weak_scaling_scene(cfg,1)
pos=slab_scene.seed_scene_particles(dim0,maxp,cubes)
# keep only the lower dam (x,z < 32): fits every grid below
pos=pos[(pos[:,0]<40)&(pos[:,2]<40)]
for dim in ((256,256,256),(512,512,512),(496,512,512),(512,496,512),(512,512,256),(256,256,1024),(1024,256,256)):
    f=blub_amd.HybridFluid(dim,len(pos)+64)
    f.set_gravity_grid(gravity); f.set_particles(pos)
    for _ in range(40): f.step(dt)
    f.synchronize(); f.profile_reset(); f.profile_enable(True)
    for _ in range(30): f.step(dt)
    f.synchronize(); pr=f.profile_read()
    print(dim, "bricks", f.brick_counts()["fluid"], {k:(v["launches"]//30, round(v["total_ms"]*1e3/v["launches"],2)) for k,v in pr.items() if k in ("pcg_iter","pcg_init","gather_velocity","build_lists","extrapolate","advect")})
    f.close()
P
