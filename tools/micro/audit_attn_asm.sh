#!/bin/bash
# Audit of attn_fwd64_kernel's asm-owned registers (fastdm_amd/csrc/attention.hip): compiles attention.hip with -save-temps and
# fails if (1) the kernel spills or uses scratch, (2) any instruction OUTSIDE the inline-asm statements touches a240..a255
# (the V^T fragment registers owned by the asm), (3) an s_load / compiler LDS instruction sits inside the tile loop (it would
# break the hand-counted lgkmcnt waits).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
TMP=$(mktemp -d)
cd "$TMP"
# FDM_AUDIT_ASM=<file>: audit that assembly (build.sh leaves the library's own at fastdm_amd/_lib/attention.gfx950.s) instead of compiling
if [ -n "$FDM_AUDIT_ASM" ]; then
  S="$FDM_AUDIT_ASM"
else
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-result -save-temps \
      -c "$HERE/../../fastdm_amd/csrc/attention.hip" -o a.o 2> err.txt || { tail -20 err.txt; exit 1; }
  S=attention-hip-amdgcn-amd-amdhsa-gfx950.s
fi
case "$S" in /*) ;; *) S="$TMP/$S" ;; esac
python3 "$HERE/audit_no_scratch.py" "$S"     # every fdm:: kernel of the file: zero private segment
# MASKED=1 audits the block-mask instantiation, HD=64 the head_dim-64 one, PV8=1 the fp8 P.V kernel (attn_fwd64_pv8_kernel: its asm
# owns a224..a255 -- four 8-register V^T fragments -- instead of a240..a255)
# QK8=1: the fp8 QK^T + fp8 P.V kernel (attn_fwd64_qk8_kernel: a192..a255 -- four K and four V^T fragments)
if [ -n "$QK8" ]; then KN="21attn_fwd64_qk8_kernel"; export FDM_AUDIT_OWNED_LO=192;
elif [ -n "$PV8" ]; then KN="21attn_fwd64_pv8_kernel"; export FDM_AUDIT_OWNED_LO=224; else KN="17attn_fwd64_kernel"; export FDM_AUDIT_OWNED_LO=240; fi
K="_ZN3fdm${KN}ILi${HD:-128}ELb${MASKED:-0}EEEvNS_10AttnParamsE"
awk -v k="^$K:" '$0 ~ k {on=1} on {print} on && /^\.Lfunc_end/ {exit}' $S > k.s
grep -A40 "amdhsa_kernel _ZN3fdm${KN}ILi${HD:-128}ELb${MASKED:-0}" $S | grep -E "next_free_vgpr|accum_offset|private_segment_fixed_size"
python3 - <<'PY'
import os, re, sys
LO = int(os.environ.get("FDM_AUDIT_OWNED_LO", "240"))
def owned(code):
    for m in re.finditer(r"\ba(\d+)\b", code):
        if int(m.group(1)) >= LO: return True
    for m in re.finditer(r"a\[(\d+):(\d+)\]", code):
        if int(m.group(2)) >= LO: return True
    return False
lines = open("k.s").read().split("\n")
inasm = False
bad = []
# the tile loop = the loop with the MFMAs in its body (the masked variant has others: it builds its tile list in two, and waves outside
# their stretch of the list keep the ring going in two more, where compiler-managed LDS reads and waits are fine)
loop_start = loop_end = None
best = 0
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", l)
    if m:
        back = [j for j in range(i, len(lines)) if re.search(r"s_c?branch\w*\s+" + re.escape(m.group(1)) + r"\b", lines[j])]
        if back:
            n = sum("v_mfma" in x for x in lines[i:back[-1]])
            if n > best:
                best, loop_start, loop_end = n, i, back[-1]
for i, l in enumerate(lines):
    if "#ASMSTART" in l: inasm = True; continue
    if "#ASMEND" in l: inasm = False; continue
    code = l.split(";")[0]
    if not inasm and owned(code):
        bad.append((i, l.strip()))
    if not inasm and loop_start is not None and loop_start < i < loop_end and re.search(r"^\s*(s_load|s_buffer_load|ds_)", code) and "s_endpgm" not in code:
        # compiler-generated scalar loads / LDS ops inside the tile loop
        bad.append((i, "in-loop: " + l.strip()))
# fp8 P.V kernel: hipcc takes an MFMA asm statement's accumulator for ready when the statement ends and does park v_accvgpr_read /
# v_accvgpr_mov copies of it right behind (seen in round 5: a tile missing in half of the registers).  A 16-pass MFMA needs ~20 wait
# states: no compiler instruction may touch the destination accumulators within the next 24 instructions (the rare-path MFMAs carry
# their own s_nops inside the statement, which count).
code_lines = [(i, l) for i, l in enumerate(lines) if re.match(r"^\s+[a-z]", l)]
pos = {i: n for n, (i, _) in enumerate(code_lines)}
for i, l in enumerate(lines):
    m = re.match(r"\s+v_mfma_f32_32x32x64_f8f6f4 a\[(\d+):(\d+)\]", l)
    if not m: continue
    lo, hi = int(m.group(1)), int(m.group(2))
    n0 = pos[i]
    waited = 0
    for j, l2 in code_lines[n0 + 1:n0 + 25]:
        nm = re.match(r"\s+s_nop (\d+)", l2)
        waited += int(nm.group(1)) + 1 if nm else 1
        if waited > 24: break
        if "v_mfma" in l2: continue
        regs = [int(x) for x in re.findall(r"\ba(\d+)\b", l2.split(";")[0])]
        for a, b in re.findall(r"a\[(\d+):(\d+)\]", l2.split(";")[0]): regs += list(range(int(a), int(b) + 1))
        if any(lo <= r <= hi for r in regs):
            bad.append((j, "accumulator a[%d:%d] touched %d wait states after its fp8 MFMA: %s" % (lo, hi, waited, l2.strip())))
            break
scratch = [l for l in lines if re.search(r"^\s*scratch_", l)]
n_ins = sum(1 for l in lines if re.match(r"^\s+[a-z]", l))
print("instructions: %d, mfma %d (%d in the tile loop, lines %s-%s), scratch ops %d" % (n_ins, sum("v_mfma" in l for l in lines), best, loop_start, loop_end, len(scratch)))
if loop_start is None: bad.append((0, "no loop with MFMAs found"))
for b in bad[:20]: print("VIOLATION line %d: %s" % b)
sys.exit(1 if bad or scratch else 0)
PY
echo "audit ok"
