"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: dispatches, and per counter the sum and the
per-dispatch mean.  Usage: pmc_summary.py <dir-or-csv> [name-filter]"""
import csv, glob, os, sys, collections
csv.field_size_limit(1 << 30)
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
paths = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in paths:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("void ", "")
            name = name.split("(")[0][:80]
            if flt and flt not in name:
                continue
            a = agg[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
print(f"{'kernel':80s} {'counter':28s} {'dispatches':>10s} {'sum':>16s} {'mean/dispatch':>16s}")
for name in sorted(agg):
    for ctr, (n, tot) in sorted(agg[name].items()):
        print(f"{name:80s} {ctr:28s} {n:10d} {tot:16.6g} {tot / n:16.6g}")
