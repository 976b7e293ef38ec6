#!/usr/bin/env python3
"""Per-kernel wave-cycle attribution from the PMC passes of tools/stall_table.sh: mean counter value per dispatch, then
   parked %   = SQ_WAIT_ANY / SQ_WAVE_CYCLES        (s_waitcnt, s_barrier)
   stalled %  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (an instruction is ready but cannot issue: MFMA dependency, busy pipe)   [LDS part separately]
   issuing %  = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, split by unit (VALU incl. MFMA issue, LDS, VMEM, scalar, misc)
   MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (the matrix pipe of a SIMD), coexec = cycles with VALU and MFMA both executing
and instructions per wave.  usage: stall_table.py <pmc dir> ..."""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
            if not name.startswith("fdm::"):
                continue
            a = agg[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
            if name.startswith("fdm::"):
                dur[name][0] += 1
                dur[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def m(name, c):
    n, tot = agg[name].get(c, [0, 0.0])
    return tot / n if n else float("nan")


rows = []
for name in agg:
    wc = m(name, "SQ_WAVE_CYCLES")
    if not wc == wc or wc <= 0:
        continue
    us = dur[name][1] / max(dur[name][0], 1)
    rows.append((us * dur[name][0], name, us, wc))
rows.sort(reverse=True)
print(f"{'kernel':58s} {'us':>7s} {'parked%':>8s} {'stall%':>7s} {'(lds%)':>7s} {'issue%':>7s} | {'valu%':>6s} {'lds%':>5s} {'vmem%':>6s} {'sca%':>5s} {'misc%':>6s} | {'MFMAbusy%':>9s} {'coexec%':>8s} | "
      f"{'VALU/wave':>9s} {'MFMA/wave':>9s} {'trans':>6s} {'cvt':>6s} {'SALU':>6s} {'LDS':>6s} {'VMEM':>6s} | {'eff.GHz':>7s}")
for _, name, us, wc in rows[:14]:
    g = lambda c: m(name, c)
    waves = g("SQ_WAVES")
    busy = g("SQ_BUSY_CYCLES")
    pct = lambda c: 100.0 * g(c) / wc
    simd = g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0         # SIMD-cycles of the dispatch (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs): the MfmaUtil expression of `rocprofv3 -L`
    mf = 100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / simd if simd == simd and simd > 0 else float("nan")
    co = 100.0 * g("SQ_VALU_MFMA_COEXEC_CYCLES") / simd if simd == simd and simd > 0 else float("nan")
    ghz = g("GRBM_GUI_ACTIVE") / 8.0 / (us * 1e3) if us > 0 else float("nan")
    pw = lambda c: g(c) / waves if waves == waves and waves > 0 else float("nan")
    print(f"{name:58s} {us:7.1f} {pct('SQ_WAIT_ANY'):8.1f} {pct('SQ_WAIT_INST_ANY'):7.1f} {pct('SQ_WAIT_INST_LDS'):7.1f} {pct('SQ_ACTIVE_INST_ANY'):7.1f} | "
          f"{pct('SQ_ACTIVE_INST_VALU'):6.1f} {pct('SQ_ACTIVE_INST_LDS'):5.1f} {pct('SQ_ACTIVE_INST_VMEM'):6.1f} {pct('SQ_ACTIVE_INST_SCA'):5.1f} {pct('SQ_ACTIVE_INST_MISC'):6.1f} | "
          f"{mf:9.1f} {co:8.1f} | {pw('SQ_INSTS_VALU'):9.0f} {pw('SQ_INSTS_MFMA'):9.0f} {pw('SQ_INSTS_VALU_TRANS_F32'):6.0f} {pw('SQ_INSTS_VALU_CVT'):6.0f} {pw('SQ_INSTS_SALU'):6.0f} "
          f"{pw('SQ_INSTS_LDS'):6.0f} {pw('SQ_INSTS_VMEM'):6.0f} | {ghz:7.2f}")
