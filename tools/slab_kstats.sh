#!/bin/bash
# rocprofv3 kernel trace of the 2-slab loopback run (tools/slab_loopback_bench.py) -> per-kernel totals
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/kts
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/kts -o t -- python $root/tools/slab_loopback_bench.py corner_dams_256 2 60 5 > $root/gpurun_out/kts.log 2>&1
cd $root
python - <<'PY'
import sqlite3, glob, collections
db = glob.glob("gpurun_out/kts/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
acc = collections.defaultdict(lambda: [0, 0.0])
for n, a, b in rows:
    n = n.split("(")[0].replace("void ", "").replace("blubk::", "")[:60]
    acc[n][0] += 1; acc[n][1] += (b - a) / 1e3
tot = sum(v[1] for v in acc.values())
print("total kernel us", round(tot), "dispatches", len(rows))
for n, (k, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-62s %7d %10.1f us  %5.1f %%  avg %.2f" % (n, k, us, 100 * us / tot, us / k))
PY
tail -2 gpurun_out/kts.log
