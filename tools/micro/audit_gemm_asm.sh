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
# gemm_p4_kernel: the body is ONE asm statement whose inputs hipcc puts into v0-v7 / s0-s7 -- registers the body itself
# overwrites (they are not on the clobber list: every other register is, so there is nowhere else for the inputs to go).  That is sound
# only while (1) each input register is read for the last time before the body first writes it, and (2) nothing but s_endpgm follows
# the statement.  Checked here on the compiled assembly for every instantiation.
python3 - "$OUT" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
kerns = re.findall(r'^(_ZN3fdm14gemm_p4_kernel\w+):', s, re.M)
assert kerns, "no gemm_p4_kernel instantiation found"
bad = 0
def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r'([sv])\[(\d+):(\d+)\]', tok)
    if m: return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r'-?\|?([sv])(\d+)\|?', tok)
    return {f"{m.group(1)}{m.group(2)}"} if m else set()
NO_DST = ("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop", "global_store", "ds_write", "s_endpgm", "s_bitcmp", "s_setprio")
for kern in kerns:
    body = re.search(re.escape(kern) + r':(.*?)\.Lfunc_end', s, re.S).group(1)
    m = re.search(r'#ASMSTART(.*?)#ASMEND(.*)', body, re.S)
    ins = [l.split(';')[0].strip() for l in m.group(1).split('\n')]
    ins = [l for l in ins if l and not l.endswith(':') and not l.startswith(('.', '//'))]
    tail = [l.split(';')[0].strip() for l in m.group(2).split('\n')]
    tail = [l for l in tail if l and not l.endswith(':') and not l.startswith(('.', ';'))]
    if [l for l in tail if not l.startswith(('s_endpgm', 's_nop', 's_waitcnt', 's_code_end'))]:
        print(f"{kern}: instructions behind the asm statement: {tail[:4]}"); bad += 1
    # the five inputs, by the instruction forms gen_gemm_p4.py emits for them
    uses = {}        # register -> index of its last input read
    def note(i, tok):
        for r in regs(tok): uses[r] = i
    found = 0
    for i, l in enumerate(ins[:400]):
        ops = [o.strip() for o in l.split(None, 1)[1].split(',')] if ' ' in l else []
        if i == 0 and l.startswith("s_mov_b64"): note(i, ops[1]); found += 1
        elif i == 1 and l.startswith("s_mov_b32"): note(i, ops[1]); found += 1
        elif i == 2 and l.startswith("v_and_b32"): note(i, ops[2]); found += 1
        elif i == 3 and l.startswith("s_mov_b32"): note(i, ops[1]); found += 1
        elif l.startswith("s_cselect_b64") and i < 60 and re.fullmatch(r's\[\d+:\d+\]', ops[1]) and regs(ops[1]) <= {f"s{k}" for k in range(8)}:
            note(i, ops[1]); found += 1
    if found != 6:
        print(f"{kern}: expected 6 input uses (4 moves + 2 zero-bias selects), recognised {found}"); bad += 1
    for i, l in enumerate(ins[:max(uses.values(), default=0)]):
        if l.startswith(NO_DST) or ' ' not in l: continue
        dst = l.split(None, 1)[1].split(',')[0]
        hit = {r for r in regs(dst) if r in uses and i < uses[r]}
        if hit:
            print(f"{kern}: input register {sorted(hit)} overwritten by instruction {i} '{l}' before its last read at {max(uses[r] for r in hit)}"); bad += 1
    print(f"{kern}: inputs in {sorted(uses)}: read before the body writes them; nothing behind the statement" if not bad else f"{kern}: FAILED")
sys.exit(1 if bad else 0)
PY

