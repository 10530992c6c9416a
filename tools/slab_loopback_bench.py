"""z-slab protocol overhead on ONE GPU (loopback transport): N stacked copies of a scene as N slabs vs the single-domain engine
on one copy.  Not a scaling measurement (all slabs share the GPU) -- it prices the protocol itself: extra kernels, halo copies,
host synchronisations and the number of transport operations a real multi-GPU run would issue per step."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd  # noqa: E402
from blub_amd import slab_scene  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(scene="corner_dams_256", slabs=2, steps=60, warmup=5, schedule="single_reduction", async_exchange=1, transport="direct"):
    dt = blub_amd.default_simulation_delta()
    path = os.path.join(ROOT, "scenes", scene + ".json")
    cfg = blub_amd.Scene.parse(path=path).config
    dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, slabs)
    pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
    # max_num_particles is PER SLAB: a quarter more than an even share (every slab holds one copy of the scene) -- particle kernels are launched
    # for the capacity once the counts live on the device, so handing every slab the capacity of the whole domain costs N^2 empty workgroups
    g = blub_amd.SlabGroup(dim, len(pos) // slabs * 5 // 4 + 65536, local=slabs)
    g.set_gravity_grid(gravity)
    g.set_transport(transport)
    g.set_pcg_schedule(schedule)
    g.set_async_exchange(bool(async_exchange))
    g.set_particles(pos)
    for _ in range(warmup):
        g.step(dt)
    g.synchronize()
    ops0, t0 = g.transport_ops(), time.perf_counter()
    for _ in range(steps):
        g.step(dt)
    el_enqueue = time.perf_counter() - t0      # the host has issued everything (ONE thread drives all N slabs here; a real group has a process per slab)
    g.synchronize()
    el = time.perf_counter() - t0
    ops = (g.transport_ops() - ops0) / steps
    syncs = g.host_syncs()
    g.close()
    s = blub_amd.Scene(path=path)
    f = s.fluid()
    f.set_pcg_schedule(schedule)
    for _ in range(warmup):
        s.step(dt)
    f.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step(dt)
    f.synchronize()
    el1 = time.perf_counter() - t0
    f.close()
    print(json.dumps({"scene": scene, "slabs_on_one_gpu": slabs, "grid": list(dim), "particles": len(pos), "steps": steps,
                      "slab_group_ms_per_step": round(el / steps * 1e3, 3), "host_enqueue_ms_per_step": round(el_enqueue / steps * 1e3, 3), "single_domain_one_copy_ms_per_step": round(el1 / steps * 1e3, 3),
                      "protocol_overhead_vs_n_sequential_copies": round(el / (slabs * el1), 3), "transport_ops_per_step": round(ops, 1),
                      "pcg_schedule": schedule, "transport": transport, "async_particle_exchange": bool(async_exchange), "host_syncs_particle_exchanges_total": syncs[0], "host_syncs_done_polls_total": syncs[1]}))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(*(a[0:1] or ["corner_dams_256"]), *[int(v) for v in a[1:4]], *(a[4:5]), *[int(v) for v in a[5:6]], *(a[6:7]))
