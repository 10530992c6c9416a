"""Development probe: durations of the K kernels of ONE solve that converges at the first check (iterations >= 6 are no-op launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt = blub_amd.default_simulation_delta()
scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_256.json"))
f = scene.fluid()
for _ in range(100):
    scene.step(dt)
f.synchronize()
print("bricks", f.brick_counts())
f.run_stage("transfer", dt); f.run_stage("divergence", dt)
f.set_solver_config(0, error_tolerance=1e9, max_num_iterations=32, error_check_frequency=4)
f.profile_enable(True); f.profile_reset()
f.run_stage("solve_velocity", dt)
f.synchronize()
ev = [e for e in f.profile_trace() if e["name"] == "pcg_iter"]
print("iterations reported:", f.solver_stats(0))
print("K durations us:", " ".join("%.1f" % e["duration_us"] for e in ev))
print("K start deltas us:", " ".join("%.1f" % (b["start_us"] - a["start_us"]) for a, b in zip(ev[:-1], ev[1:])))
