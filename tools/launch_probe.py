"""Diagnostic: does queue depth change GPU throughput?  Same steps (10..10+N) of a fresh simulation, sync every k steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd

def run(k, steps=60, inflight=None):
    scene = blub_amd.Scene(path=os.path.join(os.path.dirname(__file__), "..", "scenes", "corner_dams_256.json"))
    f = scene.fluid()
    dt = blub_amd.default_simulation_delta()
    if inflight is not None:
        f.set_max_steps_in_flight(inflight)
    for _ in range(10):
        f.step(dt)
    f.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        f.step(dt)
        if k and (i + 1) % k == 0:
            f.synchronize()
    f.synchronize()
    el = time.perf_counter() - t0
    print("sync every %3s steps, max in flight %s: %.3f ms/step" % (k if k else "inf", inflight, el / steps * 1e3), flush=True)
    f.close()

for fl in (1, 2, 3, 4, 6, 0):
    run(0, inflight=fl)
