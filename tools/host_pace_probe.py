"""Does the trajectory depend on how fast the host enqueues?  corner_dams_128, 6 steps, rebinning every 2 steps, the gather pinned:
(a) Python calls scene.step per step (a slow host), (b) the C loop of the controller (blub_controller_fast_forward_steps_fluid: as fast as a host can be),
(c) the C loop with one step in flight.  Every variant twice; all pairs compared (particles matched by position)."""
import itertools
import json
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import blub_amd  # noqa: E402
from blub_amd.simulation_controller import SimulationController  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
scene_name = sys.argv[1] if len(sys.argv) > 1 else "corner_dams_128"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
fixed = "fixed" in sys.argv[3:]      # solves of a fixed 120 iterations (no convergence decision) instead of tolerance 2e-6
tunes = dict(kv.split("=") for kv in sys.argv[3:] if "=" in kv)


def run(mode):
    sc = blub_amd.Scene(path=os.path.join(ROOT, "scenes", scene_name + ".json"))
    f = sc.fluid()
    for w in (0, 1):
        f.set_solver_config(w, error_tolerance=0.0 if fixed else 2e-6, max_num_iterations=120 if fixed else 400, error_check_frequency=8)
    f.particle_rebinning_step_frequency = 2
    for k, v in tunes.items():
        f.set_tuning(k, int(v))
    dt = blub_amd.default_simulation_delta()
    if mode == "python":
        for _ in range(steps):
            sc.step(dt)
    else:
        if mode == "c_loop_1_in_flight":
            f.set_max_steps_in_flight(1)
        ctl = SimulationController()
        assert ctl.fast_forward_steps_fluid(f, ctl.simulation_delta_ns * steps) == steps
    f.synchronize()
    f.update_statistics()
    its = [s.iteration_count for s in f.pressure_solver_stats_velocity()], [s.iteration_count for s in f.pressure_solver_stats_density()]
    pos = f.get_particles()[0][:, :3].astype(np.float64)
    f.close()
    return pos, its


res = {}
for mode in ("python", "c_loop", "c_loop_1_in_flight"):
    for k in (1, 2):
        res["%s#%d" % (mode, k)] = run(mode)
out = {"scene": scene_name, "steps": steps, "tunes": tunes, "solves": "fixed 120 iterations" if fixed else "tolerance 2e-6, check every 8", "iterations": {k: v[1] for k, v in res.items()}, "pairs": {}}
for a, b in itertools.combinations(res, 2):
    d, idx = cKDTree(res[b][0]).query(res[a][0], k=1)
    out["pairs"]["%s vs %s" % (a, b)] = {"median": float(np.median(d)), "p99.9": float(np.quantile(d, 0.999)), "max": float(d.max()), "one_to_one": bool(len(np.unique(idx)) == len(d))}
print(json.dumps(out, indent=1))
