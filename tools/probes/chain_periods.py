import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
st = np.array([r[1] for r in rows], float); en = np.array([r[2] for r in rows], float)
per = np.diff(st) / 1e3; dur = (en - st) / 1e3
# phases alternate: plain block then graph block, per (grid, variant); print quantiles over chunks of 66*205 launches
chunk = 66 * 205
for k in range(0, len(per) - chunk + 1, chunk):
    p, d = per[k:k + chunk], dur[k:k + chunk]
    print("launches %7d..: period median %.2f us (p10 %.2f, p90 %.2f), duration median %.2f us" % (k, np.median(p), np.quantile(p, 0.1), np.quantile(p, 0.9), np.median(d)))
