"""ONE domain cut into N z-slabs on ONE GPU (loopback group, strong scaling): uniform against fluid-weighted cut planes, and the memory modes of the slabs.

All slabs share the GPU and one host thread, so the wall clock of a step is the SUM over the slabs and says nothing about scaling.  What a node with
a GPU per slab would see is bounded from below by the BUSIEST slab, so this tool reports, from the library's own per-launch profile
(blub_fluid_profile_*), the GPU-busy microseconds per step of every slab: max = the critical path of a perfectly overlapped group, sum = what this one
GPU executes.  Round-4 review, item 1a: uniform cuts of the metric's scene leave six of eight slabs without fluid -- two ranks carry everything.

cuts_mode "dynamic": weighted cuts at t = 0, then blub_slab_group_rebalance every REBALANCE_EVERY (16) steps -- slabs that hold the whole grid.

usage: python tools/slab_cuts_bench.py [scene] [slabs] [steps] [warmup] [uniform|weighted|dynamic] [coarse|fine_grained|uncached] [direct|host]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd  # noqa: E402
from blub_amd import slab_scene  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(scene="corner_dams_256", slabs=8, steps=60, warmup=10, cuts_mode="weighted", memory="coarse", transport="direct"):
    dt = blub_amd.default_simulation_delta()
    cfg = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", scene + ".json")).config
    dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, 1)
    pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
    uniform = [blub_amd.SlabGroup.slab_range(dim[2], slabs, i)[0] for i in range(slabs)] + [dim[2]]
    cuts = blub_amd.SlabGroup.balanced_cuts(dim, pos, slabs)[0] if cuts_mode in ("weighted", "dynamic") else uniform
    dynamic = cuts_mode == "dynamic"
    every = int(os.environ.get("REBALANCE_EVERY", "16"))
    g = blub_amd.SlabGroup(dim, len(pos) + 64, local=slabs, cuts=cuts, memory=memory, movable_cuts=dynamic)
    moves = [0]

    def step_once(k):
        if dynamic and k % every == 0 and g.rebalance(min_layers=2):
            moves[0] += 1
        g.step(dt)
    g.set_gravity_grid(gravity)
    g.set_transport(transport)
    g.set_particles(pos)
    for k in range(warmup):
        step_once(k)
    g.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step_once(warmup + k)
    g.synchronize()
    wall = (time.perf_counter() - t0) / steps
    counts = [g.local_fluid(i).num_particles() for i in range(slabs)]
    bricks_end = [g.local_fluid(i).brick_counts()["fluid"] for i in range(slabs)]
    # instrumented pass: per-slab GPU-busy time of the same number of further steps
    fl = [g.local_fluid(i) for i in range(slabs)]
    for f in fl:
        f.profile_reset()
        f.profile_enable(True)
    for k in range(steps):
        step_once(warmup + steps + k)
    g.synchronize()
    cuts_end = g.cuts()
    busy, solve = [], []
    for f in fl:
        pr = f.profile_read()
        f.profile_enable(False)
        busy.append(sum(v["total_ms"] for v in pr.values()) * 1e3 / steps)
        solve.append(sum(v["total_ms"] for k, v in pr.items() if k.startswith("pcg")) * 1e3 / steps)
    it = fl[0].total_solver_iterations()
    g.close()
    print(json.dumps({"scene": scene, "slabs_on_one_gpu": slabs, "cuts_mode": cuts_mode, "cuts": cuts, "cuts_at_end": cuts_end, "rebalance_every": every if dynamic else None, "recuts": moves[0], "memory": memory, "transport": transport, "steps": steps, "warmup": warmup,
                      "fluid_bricks_per_slab_at_t0": blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos, cuts), "fluid_bricks_per_slab_at_end": bricks_end, "particles_per_slab_at_end": counts,
                      "wall_ms_per_step_all_slabs_on_one_gpu": round(wall * 1e3, 3),
                      "gpu_busy_us_per_step_per_slab": [round(b, 1) for b in busy], "pcg_us_per_step_per_slab": [round(b, 1) for b in solve],
                      "busiest_slab_us_per_step": round(max(busy), 1), "sum_of_slabs_us_per_step": round(sum(busy), 1), "imbalance_max_over_mean": round(max(busy) / (sum(busy) / slabs), 3),
                      "total_solver_iterations_slab0": it}))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(*(a[0:1] or ["corner_dams_256"]), *[int(v) for v in a[1:4]], *a[4:7])
