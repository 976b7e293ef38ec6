"""profiles/pmc_traffic.json from the two pmc_summary.py outputs (FETCH_SIZE pass, WRITE_SIZE pass).
rocprofv3 reports both in KiB-ish units of 1 KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is
doubled (MI355X_MICROARCH.md, HBM section).  Usage: pmc_to_json.py <fetch.txt> <write.txt> <out.json> <note>"""
import json, os, sys, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def parse(path):
    d = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"(.{80}) (\S+)\s+(\d+)\s+(\S+)\s+(\S+)$", line)
        if m:
            d[m.group(1).strip()] = (int(m.group(3)), float(m.group(5)))
    return d
f, w = parse(sys.argv[1]), parse(sys.argv[2])
from fastdm_amd import _native
# the digest of the kernel sources these counters were taken with: bench.py reports `traffic` only while it still matches
out = {"note": sys.argv[4], "srcdigest": _native.current_source_digest(), "kernels": {}}
for k in f:
    if k in w:
        rd, wr = 2.0 * f[k][1] * 1024.0, w[k][1] * 1024.0
        out["kernels"][k] = {"dispatches": f[k][0], "fetch_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
