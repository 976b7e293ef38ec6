"""MFMA / LDS utilisation per kernel from the gpu_evidence_mfma.sh passes.
rocprofv3 reports each counter summed over the 8 XCDs (GRBM_GUI_ACTIVE) or over all SIMDs (SQ_*): per launch,
  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)      (the MfmaUtil expression of `rocprofv3 -L`)
(GRBM_GUI_ACTIVE / 8 / duration is NOT printed as a clock: it reads high on dispatches shorter than ~0.3 ms -- 5 GHz for a
7 us kernel, 2.0-2.3 GHz for GEMMs whose in-kernel stamps say 1.35-1.4 GHz -- MI355X_MICROARCH "DVFS give-back".)
Usage: pmc_util.py <dir with <tag>_pmc_*.txt> <tag> <kernel_stats.csv>"""
import csv, re, sys, collections
d, tag, stats = sys.argv[1], sys.argv[2], sys.argv[3]
vals = collections.defaultdict(dict)
for grp in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
    for line in open(f"{d}/{tag}_pmc_{grp}.txt").read().splitlines()[1:]:
        m = re.match(r"(.{80}) (\S+)\s+(\d+)\s+(\S+)\s+(\S+)$", line)
        if m:
            vals[m.group(1).strip()][m.group(2)] = float(m.group(5))
dur = {}
for r in csv.DictReader(open(stats)):
    dur[r["Name"].replace("void ", "").split("(")[0][:80]] = float(r["AverageNs"]) * 1e-9
print(f"{'kernel':52s} {'avg us':>8s} {'MFMA util %':>11s} {'VALU issue %':>12s} {'LDS issue %':>11s} {'LDS conflict %':>14s}")
for k, v in sorted(vals.items()):
    if "GRBM_GUI_ACTIVE" not in v or k not in dur:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    simd = cyc * 1024.0
    print(f"{k[:52]:52s} {dur[k] * 1e6:8.1f} {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / simd:11.1f} "
          f"{100 * v.get('SQ_ACTIVE_INST_VALU', 0) / simd:12.1f} {100 * v.get('SQ_ACTIVE_INST_LDS', 0) / simd:11.1f} "
          f"{100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):14.1f}")
