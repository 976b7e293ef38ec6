#!/bin/bash
# gemm_w4_kernel keeps its 256 accumulator registers in a[0:255], touched by inline asm only.  This audit compiles gemm.hip to
# assembly and fails if hipcc itself references an AGPR or scratch memory in any gemm_w4_kernel instantiation (a compiler spill
# into a[..] would silently corrupt accumulators).  Usage: tools/micro/audit_gemm_asm.sh [extra hipcc flags]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
# FDM_AUDIT_ASM=<file>: audit that assembly (build.sh leaves the library's own at fastdm_amd/_lib/gemm.gfx950.s) instead of compiling
if [ -n "$FDM_AUDIT_ASM" ]; then
  OUT="$FDM_AUDIT_ASM"
else
  OUT="${TMPDIR:-/tmp}/fdm_gemm_audit.s"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -S --cuda-device-only \
    "$HERE/../../fastdm_amd/csrc/gemm.hip" -o "$OUT" 2>/dev/null
fi
python3 "$HERE/audit_no_scratch.py" "$OUT"     # every fdm:: kernel of the file: zero private segment
python3 - "$OUT" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
bad_total = 0
kerns = re.findall(r'^(_ZN3fdm14gemm_w4_kernel\w+):', s, re.M)
assert kerns, "no gemm_w4_kernel instantiation found"
for kern in kerns:
    body = re.search(re.escape(kern) + r':(.*?)\.Lfunc_end', s, re.S).group(1)
    inasm, bad = False, []
    for l in body.split('\n'):
        if '#ASMSTART' in l: inasm = True; continue
        if '#ASMEND' in l: inasm = False; continue
        if not inasm and (re.search(r'\ba\[?\d+', l) or 'accvgpr' in l or 'scratch_' in l or 's_swappc' in l): bad.append(l.strip())   # a real call (an un-inlined lambda) clobbers by ABI and brings scratch
    # hipcc's own LDS-DMAs must load M0 themselves (the asm DMAs leave their value in it)
    code, flags = [], []
    inasm = False
    for l in body.split('\n'):
        t_ = l.strip()
        if '#ASMSTART' in t_: inasm = True; continue
        if '#ASMEND' in t_: inasm = False; continue
        if not t_ or t_.startswith(';') or t_.startswith('.'): continue
        code.append(t_); flags.append(inasm)
    for i, l in enumerate(code):
        if l.startswith('global_load_lds') and not flags[i]:
            j = i - 1
            ok = False
            while j >= 0 and i - j <= 12:
                if re.match(r's_\w+ m0,', code[j]) and ' m0, m0' not in code[j] and not flags[j]: ok = True; break
                if flags[j] or code[j].startswith('global_load_lds'): break
                j -= 1
            if not ok: bad.append('compiler LDS-DMA without its own M0 write: ' + l)
    meta = re.search(r'\.name:\s+' + re.escape(kern) + r'\n(.*?)\.wavefront_size', s, re.S)
    print(f"{kern}: {len(bad)} compiler AGPR / scratch references")
    for b in bad[:8]: print("   ", b)
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
PY
