import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import blub_amd
from blub_amd import slab_scene
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dt = blub_amd.default_simulation_delta()
cfg = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", "corner_dams_256.json")).config
dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, 2)
pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
g = blub_amd.SlabGroup(dim, len(pos) + 64, local=2)
g.set_gravity_grid(gravity); g.set_particles(pos)
for _ in range(20): g.step(dt)
g.synchronize()
f0 = g.local_fluid(0)
f0.profile_enable(True); f0.profile_reset()
t0 = time.perf_counter()
for _ in range(10): g.step(dt)
g.synchronize()
el = time.perf_counter() - t0
prof = f0.profile_read()
print("wall ms/step (profiled)", el / 10 * 1e3)
print({k: round(v["total_ms"] / 10 * 1e3, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
print({k: round(v["launches"] / 10, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
print("sum us/step slab0:", round(sum(v["total_ms"] for v in prof.values()) / 10 * 1e3, 1))
