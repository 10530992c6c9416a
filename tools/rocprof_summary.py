"""Turns a rocprofv3 (rocpd sqlite) result into the per-kernel stats table committed under profiles/."""
import sqlite3
import sys

import numpy as np


def main(db_path, out_path=None, steps_marker="k_correct"):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    names = [r[0] for r in rows]
    st = np.array([r[1] for r in rows], float)
    en = np.array([r[2] for r in rows], float)
    nsteps = max(1, sum(steps_marker in n for n in names))
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s): %d dispatches, %d simulation steps" % (db_path.split("/")[-1], len(rows), nsteps),
             "kernel,calls,total_us,avg_us,min_us,max_us,percent,calls_per_step,us_per_step"]
    dur = (en - st) / 1e3
    tot = dur.sum()
    agg = {}
    for n, d in zip(names, dur):
        key = n.split("(")[0].replace("void ", "").replace("blubk::", "")
        agg.setdefault(key, []).append(d)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = np.array(v)
        lines.append("%s,%d,%.1f,%.2f,%.2f,%.2f,%.2f,%.1f,%.1f" % (k, len(v), v.sum(), v.mean(), v.min(), v.max(), 100 * v.sum() / tot, len(v) / nsteps, v.sum() / nsteps))
    idx = [i for i, n in enumerate(names) if steps_marker in n]
    if len(idx) > 2:
        per = np.diff(st[idx]) / 1e3
        busy = [dur[a:b].sum() for a, b in zip(idx[:-1], idx[1:])]
        lines.append("# step period us: mean %.1f median %.1f; GPU busy per step us: mean %.1f (%.1f %% of the period)" % (per.mean(), np.median(per), np.mean(busy), 100 * np.mean(busy) / per.mean()))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
