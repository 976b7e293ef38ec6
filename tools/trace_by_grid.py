#!/usr/bin/env python3
"""Group the dispatches of a rocprofv3 kernel trace by (kernel, grid size): launches, mean us, total ms -- the model-agnostic cousin of
tools/gemm_by_shape.py (a tile GEMM's grid is its tile count: launches of a few dozen tiles that take long show up at once).
    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py --model sd35 --steps 3 --warmup 1 --no-cpu-baseline --no-device-state
    python3 tools/trace_by_grid.py DIR/.../run_kernel_trace.csv [min_total_ms]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
nk = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
gk = [k for k in rows[0] if k.lower().startswith("grid_size")]
wk = [k for k in rows[0] if k.lower().startswith("workgroup_size")]
acc = defaultdict(list)
for r in rows:
    name = r[nk].replace("void ", "").split("(")[0][:70]
    grid = 1
    for k in gk[:3]:
        grid *= max(1, int(r[k] or 1))
    wg = 1
    for k in wk[:3]:
        wg *= max(1, int(r[k] or 1))
    acc[(name, grid // max(wg, 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values()) / 1e3
print(f"{'kernel':70s} {'blocks':>8s} {'launches':>8s} {'mean us':>9s} {'total ms':>9s} {'share':>6s}")
for (name, blocks), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    t = sum(v) / 1e3
    if t >= min_ms:
        print(f"{name:70s} {blocks:8d} {len(v):8d} {sum(v) / len(v):9.1f} {t:9.2f} {t / tot:6.1%}")
