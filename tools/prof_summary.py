"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: per-kernel ms per forward pass."""
import csv, sys
path, passes = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = 0.0
print(f"{'kernel':70s} {'calls/pass':>10s} {'avg_us':>10s} {'ms/pass':>9s}")
for r in rows:
    name = r["Name"].replace("void ", "")[:70]
    ms = float(r["TotalDurationNs"]) / 1e6 / passes
    if "synth_fill" in name or "copyBuffer" in name:
        continue  # weight generation / H2D at load time
    tot += ms
    if ms >= 0.05:
        print(f"{name:70s} {float(r['Calls'])/passes:10.1f} {float(r['AverageNs'])/1e3:10.1f} {ms:9.3f}")
print(f"{'TOTAL GPU busy per pass (excl. load-time kernels)':70s} {'':10s} {'':10s} {tot:9.3f}")
