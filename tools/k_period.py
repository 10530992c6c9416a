"""Start-to-start period of consecutive PCG iteration kernels in a rocprofv3 kernel trace (launch gap included)."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
names = [r[0] for r in rows]; st = np.array([r[1] for r in rows], float); en = np.array([r[2] for r in rows], float)
sel = [i for i, n in enumerate(names) if "k_pcg1_iter_s" in n]
per, dur, gap = [], [], []
for a, b in zip(sel[:-1], sel[1:]):
    if b == a + 1 and en[a] - st[a] > 3000 and en[b] - st[b] > 3000:
        per.append(st[b] - st[a]); dur.append(en[a] - st[a]); gap.append(st[b] - en[a])
per, dur, gap = np.array(per) / 1e3, np.array(dur) / 1e3, np.array(gap) / 1e3
print("active K kernels: n=%d  period mean %.2f us (median %.2f)  duration mean %.2f  gap mean %.2f (median %.2f)" % (len(per), per.mean(), np.median(per), dur.mean(), gap.mean(), np.median(gap)))
noop = [i for i in sel if en[i] - st[i] < 3000]
g2 = [st[i + 1] - st[i] for i in noop[:-1] if i + 1 in set(noop)]
if g2:
    print("no-op K kernels: n=%d period mean %.2f us" % (len(g2), np.mean(g2) / 1e3))
d_all = (en[sel] - st[sel]) / 1e3
hist, edges = np.histogram(d_all, bins=[0, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5, 9.5, 10.5, 12, 14, 16, 20, 30])
print("K duration histogram (us):", ", ".join("%g-%g: %d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))
