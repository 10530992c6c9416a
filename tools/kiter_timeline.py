"""Intra-kernel timeline of K(i), the kernel that dominates the headline step (round-5 review, item 3c): workgroup 0's time stamps at its phase boundaries
(blub_fluid_read_phase_stamps), averaged over the iterations and solves of a window of the headline scene.
usage: python tools/kiter_timeline.py [scene] [warmup] [steps]  ->  JSON"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import blub_amd  # noqa: E402

scene_name = sys.argv[1] if len(sys.argv) > 1 else "corner_dams_256"
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dt = blub_amd.default_simulation_delta()
sc = blub_amd.Scene(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes", scene_name + ".json"))
f = sc.fluid()
f.set_tuning("pcg_phase_stamps", 1)
for _ in range(warmup):
    sc.step(dt)
f.synchronize()
names = ["entry -> round trip 1 back (done, list length, scalars, partials requested)", "-> partials reduced, alpha / beta known (2 barriers; descriptors in flight)",
         "-> first brick: descriptors + fields arrived, r / u / q / d / p updated, tile written", "-> tile barrier", "-> w = A u from the tile, stored", "-> block reduction, partial stored (end)"]
rows = []
for _ in range(steps):
    sc.step(dt)
    for w in (0, 1):
        st = f.phase_stamps(w).astype(np.int64)
        it = f.solver_stats(w)[1]
        for i in range(1, min(int(it) + 1, 64)):      # K(1) .. K(iterations): launches that did the work of an iteration (K(0) has no previous scalars)
            s = st[i]
            if s[0] == 0 or s[6] == 0:
                continue
            rows.append(np.diff(s[:7]) * 10.0)          # ns
rows = np.array(rows, np.float64)
mean, med = rows.mean(0), np.median(rows, 0)
out = {"scene": scene_name, "window": "%d + %d steps" % (warmup, steps), "launches_sampled": int(len(rows)), "clock": "s_memrealtime, 10 ns ticks; workgroup 0 of every K(i), i >= 1",
       "phases": [{"phase": n, "mean_ns": round(float(a), 1), "median_ns": round(float(b), 1)} for n, a, b in zip(names, mean, med)],
       "workgroup_0_entry_to_end_mean_ns": round(float(rows.sum(1).mean()), 1)}
print(json.dumps(out))
