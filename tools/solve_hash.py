"""Development probe: bit-level fingerprint of one fixed solve (marker + right-hand side from a seeded synthetic field), to compare
two builds of the library: python tools/solve_hash.py [state.npz]  (creates the state file when it does not exist)."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/solve_state.npz"
dt = blub_amd.default_simulation_delta()
if not os.path.exists(path):
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_256.json"))
    f = scene.fluid()
    for _ in range(30):
        scene.step(dt)
    f.run_stage("transfer", dt); f.run_stage("divergence", dt)
    np.savez(path, marker=f.read_volume("marker"), b=f.read_volume("residual"))
    f.close()
s = np.load(path)
dim = s["marker"].shape[::-1]
for k in (1, 7, 32):
    h = blub_amd.HybridFluid(tuple(int(v) for v in dim), 8, binning="off")
    h.write_volume("marker", s["marker"]); h.write_volume("residual", s["b"])
    h.set_solver_config(0, error_tolerance=0.0, max_num_iterations=k, error_check_frequency=4)
    h.run_stage("solve_velocity", dt)
    p, r = h.read_volume("pressure_velocity"), h.read_volume("residual")
    fl = s["marker"] == 1
    print(k, hashlib.sha1(p[fl].tobytes()).hexdigest()[:16], hashlib.sha1(r[fl].tobytes()).hexdigest()[:16], h.solver_stats(0))
    h.close()
