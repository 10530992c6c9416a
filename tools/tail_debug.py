"""Development probe: single-reduction solve with the persistent tail forced from iteration 1 against the fully launched solve (same library,
same state): where do they differ?"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import blub_amd
    s = np.load("/tmp/solve_state.npz")
    dim = s["marker"].shape[::-1]
    dt = blub_amd.default_simulation_delta()
    out = {}
    for k in (2, 3, 6, 14):
        h = blub_amd.HybridFluid(tuple(int(v) for v in dim), 8, binning="off")
        h.write_volume("marker", s["marker"]); h.write_volume("residual", s["b"])
        h.set_solver_config(0, error_tolerance=0.0, max_num_iterations=k, error_check_frequency=4)
        h.run_stage("solve_velocity", dt)
        out["p%d" % k] = h.read_volume("pressure_velocity"); out["r%d" % k] = h.read_volume("residual"); out["s%d" % k] = h.read_volume("search")
        out["st%d" % k] = np.array(h.solver_stats(0))
        h.close()
    np.savez(sys.argv[2], **out)
    sys.exit(0)
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.exists("/tmp/solve_state.npz"):
    dt = blub_amd.default_simulation_delta()
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_128.json"))
    f = scene.fluid()
    for _ in range(30):
        scene.step(dt)
    f.run_stage("transfer", dt); f.run_stage("divergence", dt)
    np.savez("/tmp/solve_state.npz", marker=f.read_volume("marker"), b=f.read_volume("residual"))
    f.close()
for name, env in (("full", {}), ("tail", {"BLUB_PCG_TAIL_FIRST": "1"})):
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/out_%s.npz" % name], env=e)
a, b = np.load("/tmp/out_full.npz"), np.load("/tmp/out_tail.npz")
fl = np.load("/tmp/solve_state.npz")["marker"] == 1
for k in (2, 3, 6, 14):
    for v in "prs":
        x, y = a["%s%d" % (v, k)], b["%s%d" % (v, k)]
        d = np.abs(x - y) * fl
        idx = np.argwhere(d > 1e-4 * np.abs(x[fl]).max())
        print(k, v, "max diff %.3g (scale %.3g) cells %d" % (d.max(), np.abs(x[fl]).max(), len(idx)), idx[:6].tolist(), a["st%d" % k], b["st%d" % k])
