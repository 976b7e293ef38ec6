#!/bin/bash
# Audit of gemm_fp8_w4_kernel's asm-owned registers (fastdm_amd/csrc/gemm.hip): fails if (1) a w4 kernel spills / uses scratch,
# (2) between the first and the last MFMA an instruction OUTSIDE the inline-asm statements touches v128..v255 or moves an accumulator, (3) a compiler LDS
# instruction or scalar load sits inside the K loop (it would break the hand-counted lgkmcnt waits).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
TMP=$(mktemp -d)
cd "$TMP"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -save-temps \
    -c "$HERE/../../fastdm_amd/csrc/gemm.hip" -o g.o 2> err.txt || { tail -20 err.txt; exit 1; }
S=gemm-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - "$S" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
rc = 0
for name in re.findall(r"^(_ZN3fdm18gemm_fp8_w4_kernelILb[01]EEEvNS_9GemmGroupE):", txt, re.M):
    body = txt[txt.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    lines = body.split("\n")
    inasm, bad, in_loop = False, [], False
    first_mfma = last_mfma = None
    for i, l in enumerate(lines):
        if "v_mfma" in l:
            first_mfma = i if first_mfma is None else first_mfma
            last_mfma = i
    for i, l in enumerate(lines):
        if "#ASMSTART" in l: inasm = True; continue
        if "#ASMEND" in l: inasm = False; continue
        code = l.split(";")[0]
        if inasm or not code.strip() or code.strip().startswith("."): continue
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", code)] + [int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", code)]
        in_k_loop = first_mfma is not None and first_mfma <= i <= last_mfma
        if in_k_loop and any(r >= 128 for r in regs):
            bad.append((i, l.strip()))
        if re.search(r"\ba\d+\b|a\[\d+:\d+\]", code):
            bad.append((i, "compiler instruction touches an accumulation register: " + l.strip()))
        if first_mfma is not None and first_mfma <= i <= last_mfma and re.search(r"^\s*(s_load|s_buffer_load|ds_)", code):
            bad.append((i, "in-loop: " + l.strip()))
    scratch = [l for l in lines if re.search(r"^\s*scratch_", l)]
    n_ins = sum(1 for l in lines if re.match(r"^\s+[a-z]", l))
    print("%s: instructions %d, mfma %d, scratch ops %d, violations %d" % (name, n_ins, sum("v_mfma" in l for l in lines), len(scratch), len(bad)))
    for b in bad[:12]: print("  VIOLATION line %d: %s" % b)
    rc |= 1 if (bad or scratch) else 0
sys.exit(rc)
PY
echo "audit ok"
