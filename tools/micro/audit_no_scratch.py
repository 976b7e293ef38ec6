"""Every fdm:: kernel of a device-assembly file must have a zero private segment: no scratch, no spilled register.
Usage: audit_no_scratch.py <file.s>   (tools/micro/audit_{gemm,attn}_asm.sh and fastdm_amd/csrc/build.sh call it)"""
import re
import sys

s = open(sys.argv[1]).read()
bad, n = [], 0
for m in re.finditer(r"\.name:\s+(_ZN3fdm\w+)\n(.*?)\.wavefront_size", s, re.S):
    name, meta = m.group(1), m.group(2)
    if name.endswith(".kd"):
        continue
    n += 1
    seg = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
    spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", meta)
    sspill = re.search(r"\.sgpr_spill_count:\s+(\d+)", meta)
    if sspill and int(sspill.group(1)):
        print(f"   note {name}: {sspill.group(1)} SGPRs parked in VGPR lanes (v_writelane / v_readlane: no memory traffic)")
    if seg or (spill and int(spill.group(1))):
        bad.append(f"{name}: private segment {seg} B, vgpr spills {spill.group(1) if spill else '?'}, sgpr spills {sspill.group(1) if sspill else '?'}")
assert n, f"no fdm:: kernel metadata in {sys.argv[1]}"
print(f"{sys.argv[1]}: {n} fdm:: kernels, {len(bad)} with scratch")
for b in bad:
    print("   SCRATCH", b)
sys.exit(1 if bad else 0)
