"""Development probe: is the headline window bound by the host's enqueue rate or by the GPU?  Time until the last step call returns vs time until
the stream is idle (the engine lets the host run ahead by a bounded number of steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt = blub_amd.default_simulation_delta()
scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_256.json"))
f = scene.fluid()
for _ in range(10):
    scene.step(dt)
f.synchronize()
t0 = time.perf_counter()
for _ in range(120):
    scene.step(dt)
t1 = time.perf_counter()
f.synchronize()
t2 = time.perf_counter()
print("120 steps: enqueue returned after %.1f ms, stream idle after %.1f ms (%.0f steps/s); host share %.0f %%" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, 120 / (t2 - t0), 100 * (t1 - t0) / (t2 - t0)))
from blub_amd.simulation_controller import SimulationController
c = SimulationController()
n = c.fast_forward_steps_fluid(f, 120 * c.simulation_delta_ns)
print("native fast-forward of %d more steps: %.0f steps/s" % (n, n / c.computation_time_last_fast_forward))
