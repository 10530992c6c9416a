"""Development probe: run-to-run spread of the mean particle position after 12 steps of corner_dams_256, per PCG schedule (the gathers'
list order is an atomic-insertion order, so two runs of the SAME build differ in rounding and the unconverged density solve amplifies it)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt = blub_amd.default_simulation_delta()
for sched in ("reference", "single_reduction"):
    ys = []
    for rep in range(5):
        scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_256.json"))
        f = scene.fluid(); f.set_pcg_schedule(sched)
        for _ in range(12):
            scene.step(dt)
        f.synchronize()
        ys.append(f.get_particles()[0][:, :3].astype(np.float64).mean(0)); f.close()
    ys = np.array(ys)
    print(sched, "mean y per run:", np.round(ys[:, 1], 5), "spread (max-min) per axis:", np.round(ys.max(0) - ys.min(0), 5))
