"""Per-kernel sums of the counters in a rocprofv3 --pmc --output-format csv run (counter_collection.csv)."""
import collections
import csv
import glob
import sys


def main(directory):
    files = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in files:
        seen = set()
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0][:48].replace(",", ";")
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row.get("Dispatch_Id"), name)
            if key not in seen:
                seen.add(key)
                calls[name] += 1
    counters = sorted({c for v in acc.values() for c in v})
    print("kernel,calls," + ",".join(counters))
    for name, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        print(name + "," + str(calls[name]) + "," + ",".join("%.4g" % v.get(c, 0) for c in counters))


if __name__ == "__main__":
    main(sys.argv[1])
