"""Development probe: what does a launch of the PCG iteration kernel cost once the solve is finished?  Wall clock of one solve that
converges at the first check with 33 launches (27 of them after `done`) against the same solve limited to 6 launches."""
import os, sys, time
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
b = f.read_volume("residual")
def timed(maxit, tol, reps=40):
    f.set_solver_config(0, error_tolerance=tol, max_num_iterations=maxit, error_check_frequency=4)
    ts = []
    for r in range(reps):
        f.write_volume("residual", b); f.mark_pressure_initialised(0, False); f.synchronize()
        t0 = time.perf_counter(); f.run_stage("solve_velocity", dt); f.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 4] * 1e6, f.solver_stats(0)
a, sa = timed(32, 1e9)      # converges at check 4: K(0..5) active, K(6..32) find `done`
c, sc = timed(5, 1e9)       # the same work, nothing launched afterwards
d, sd = timed(32, 0.0)      # 33 active iterations
print("solve with 6 active + 27 finished launches: %.1f us %s" % (a, sa))
print("solve with 6 active launches only         : %.1f us %s" % (c, sc))
print("solve with 33 active launches             : %.1f us %s" % (d, sd))
print("=> per finished launch %.2f us, per active launch %.2f us" % ((a - c) / 27, (d - c) / 27))
