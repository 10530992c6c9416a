#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
python - <<P
import blub_amd, os, json
from blub_amd import slab_scene
dt=blub_amd.default_simulation_delta()
def run(scene, slabs, transport):
    cfg=blub_amd.Scene.parse(path="scenes/%s.json"%scene).config
    dim,scale,gravity,cubes,maxp=slab_scene.weak_scaling_scene(cfg,1)
    pos=slab_scene.seed_scene_particles(dim,maxp,cubes)
    g=blub_amd.SlabGroup(dim,len(pos)+64,local=slabs)
    g.set_gravity_grid(gravity); g.set_transport(transport); g.set_particles(pos)
    for _ in range(40): g.step(dt)
    g.synchronize()
    fl=[g.local_fluid(i) for i in range(slabs)]
    for f in fl: f.profile_reset(); f.profile_enable(True)
    for _ in range(30): g.step(dt)
    g.synchronize()
    for i in (0, slabs//2):
        pr=fl[i].profile_read()
        print(scene, slabs, transport, "slab", i, "bricks", fl[i].brick_counts()["fluid"], {k:(v["launches"]//30, round(v["total_ms"]*1e3/v["launches"],2)) for k,v in pr.items() if k.startswith("pcg")})
    g.close()
for tr in ("direct","host"):
    run("corner_dams_512", 8, tr)
run("corner_dams_256", 8, "direct")
run("corner_dams_512", 2, "direct")
sc=blub_amd.Scene(path="scenes/corner_dams_512.json"); f=sc.fluid()
for _ in range(40): sc.step(dt)
f.synchronize(); f.profile_reset(); f.profile_enable(True)
for _ in range(30): sc.step(dt)
f.synchronize(); pr=f.profile_read()
print("single corner_dams_512 bricks", f.brick_counts()["fluid"], {k:(v["launches"]//30, round(v["total_ms"]*1e3/v["launches"],2)) for k,v in pr.items() if k.startswith("pcg")})
P
