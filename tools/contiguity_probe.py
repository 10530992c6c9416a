"""Development probe: how contiguous are the P2G lists in memory as the particle order decays?  Fraction of list links (x component,
next pointer in pos.w) that point to the previous particle index, and the fraction of 64-byte position groups a list walk re-uses."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blub_amd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt = blub_amd.default_simulation_delta()
scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_256.json"))
f = scene.fluid()
for s in range(0, 121):
    if s in (1, 2, 4, 8, 15, 30, 45, 59, 61, 75, 119):
        f.run_stage("transfer", dt)          # builds the lists of this step (pos.w = next of the x list)
        p = f.get_particles()[0]
        nxt = p.view(np.uint32)[:, 3].astype(np.int64)
        idx = np.arange(len(nxt))
        valid = nxt != 0xFFFFFFFF
        contig = (nxt[valid] == idx[valid] - 1).mean()
        same_group = ((nxt[valid] >> 2) == (idx[valid] >> 2)).mean()
        print("step %3d: links %d, next == self-1: %.3f, next in the same 64 B group: %.3f" % (s, valid.sum(), contig, same_group))
        # (run_stage does not advance the simulation)
    scene.step(dt)
