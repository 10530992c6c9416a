"""Idle time between kernels of a rocprofv3 kernel trace (rocpd sqlite): which kernels are FOLLOWED by the longest gaps.
usage: trace_gaps.py DB [first_marker last_marker]"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
names = [r[0].split("(")[0].replace("void ", "").replace("blubk::", "") for r in rows]
st = np.array([r[1] for r in rows], float) / 1e3
en = np.array([r[2] for r in rows], float) / 1e3
mark = sys.argv[2] if len(sys.argv) > 2 else "k_slab_"
idx = [i for i, n in enumerate(names) if mark in n]
a, b = idx[0], idx[-1]
span = en[b] - st[a]
busy = (en[a:b + 1] - st[a:b + 1]).sum()
print("kernels %d..%d: span %.1f us, busy %.1f us (%.1f %%), %d dispatches" % (a, b, span, busy, 100 * busy / span, b - a + 1))
gaps = defaultdict(list)
for i in range(a, b):
    gaps[(names[i], names[i + 1])].append(max(0.0, st[i + 1] - en[i]))
tot = sorted(((sum(v), len(v), k) for k, v in gaps.items()), reverse=True)
for s, n, k in tot[:18]:
    print("%9.1f us in %5d gaps (avg %6.2f)  %s -> %s" % (s, n, s / n, k[0][:46], k[1][:46]))
