"""Per-step time of the P2G gather over the headline window with the gather form FORCED (rocprofv3 kernel traces: tools/gather_crossover.sh), against the fill of
the FLUID bricks: where does the list-centric form (walk + finishing kernel) stop paying against the compacting tile kernel?  The engine's rule is a threshold
on particles per FLUID brick (gather_forms, blub_fluid.hip)."""
import json
import sqlite3
import sys

import numpy as np


def per_step(db_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    steps, acc = [], 0.0
    for name, st, en in rows:
        if "k_gather_velocity3" in name or "k_gather_finish3" in name:
            acc += (en - st) / 1e3
        if "k_correct" in name:
            steps.append(acc)
            acc = 0.0
    return np.array(steps)


a, b = per_step(sys.argv[1]), per_step(sys.argv[2])      # forced list-centric (p2g_compact=0), forced compacting (p2g_compact=1)
n = min(len(a), len(b))
a, b = a[:n], b[:n]
out = {"steps": n, "list_centric_us": [round(x, 1) for x in a], "compacting_us": [round(x, 1) for x in b]}
better = a < b
out["list_centric_wins_in_steps"] = [int(i) for i in np.nonzero(better)[0]]
out["mean_us"] = {"list_centric": round(float(a.mean()), 1), "compacting": round(float(b.mean()), 1), "best_of_both_per_step": round(float(np.minimum(a, b).mean()), 1)}
print(json.dumps(out))
