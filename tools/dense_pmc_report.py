"""FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes) of the dense PCG micro-benchmark -> bytes per launch next to the
algorithmic bytes and the kernel-trace durations.  Counter unit KiB; FETCH_SIZE doubled (gfx950: wide coalesced streaming reads
are reported at 1/2, MI355X_MICROARCH.md "HBM")."""
import collections
import csv
import glob
import json
import sys


def counter_per_launch(directory, counter):
    acc, calls = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("blubk::", "")
            acc[name] += float(row["Counter_Value"])
            calls[name].add(row.get("Dispatch_Id"))
    return {k: (acc[k] * 1024.0 / max(1, len(calls[k])), len(calls[k])) for k in acc}


def main(size, d, stats_csv, out):
    n = int(size)
    N, F = n ** 3, (n - 2) ** 3
    alg = {"k_pcg_update_z": N + 20 * F, "k_pcg_dir_z": N + 12 * F}
    fetch, write = counter_per_launch(d + "/FETCH_SIZE", "FETCH_SIZE"), counter_per_launch(d + "/WRITE_SIZE", "WRITE_SIZE")
    dur = {}
    for line in open(stats_csv):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        parts = line.rsplit(",", 8)
        dur[parts[0]] = (float(parts[3]), int(parts[1]))
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blub_amd.build import source_hash
    res = {"kernel_source_sha16": source_hash(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --dense-only --dense-size %d (tools/dense_pmc.sh)" % n,
           "unit": "bytes per launch; FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section), counters are KiB"}
    lines = ["# dense %d^3 PCG micro-benchmark: N = %d cells, F = %d FLUID; traffic = 2 x FETCH_SIZE + WRITE_SIZE per launch" % (n, N, F),
             "kernel                         launches  avg_us  FETCH raw [MB]  x2 [MB]  WRITE [MB]  traffic [MB]  algorithmic [MB]  traffic/alg  alg GB/s  frac of 8 TB/s"]
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        base = k.split("<")[0]
        fr, nl = fetch[k]
        wr = write.get(k, (0.0, 0))[0]
        tr = 2 * fr + wr
        us = dur.get(k, (float("nan"), 0))[0]
        args = [v.strip() for v in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []
        a = alg.get(base) if (base == "k_pcg_update_z" or (base == "k_pcg_dir_z" and len(args) > 1 and args[1] == "false")) else None   # (dir<.., true, ..> = iteration 0: no r, no store)
        lines.append("%-30s %6d  %7.2f  %10.1f  %9.1f  %9.1f  %10.1f  %14s  %10s  %8s  %s" % (
            k[:30], nl, us, fr / 1e6, 2 * fr / 1e6, wr / 1e6, tr / 1e6, "%.1f" % (a / 1e6) if a else "--", "%.3f" % (tr / a) if a else "--",
            "%.0f" % (a / us / 1e3) if a else "--", "%.3f" % (a / us / 1e3 / 8000.0) if a else "--"))
        if a:
            key = "pcg_update" if base == "k_pcg_update_z" else "pcg_dir"
            res[key] = {"kernel": k, "fetch_raw": fr, "fetch_corrected": 2 * fr, "write": wr, "traffic": tr, "algorithmic": a, "avg_us_kernel_trace": us, "launches": nl}
    open(out + ".json", "w").write(json.dumps(res, indent=1) + "\n")
    open(out + ".txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:5])
