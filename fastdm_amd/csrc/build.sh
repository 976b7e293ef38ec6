#!/bin/bash
# Build libfastdm_amd.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $*"
# gemm_p4_kernel's body is generated assembly; the committed gemm_p4_body_*.inc are regenerated when python3 is there (same bytes)
if command -v python3 > /dev/null 2>&1; then python3 "$HERE/gen_gemm_p4.py" > /dev/null; fi
pids=()
# attention.hip: the hand-placed MFMA gaps of attn_fwd64_kernel must not be re-packed by the SLP vectoriser (it merges the
# two q-blocks' row-sum chains into v_pk_add_f32 and sinks them below the tile)
# gemm.hip / attention.hip: keep the device assembly of exactly this build beside the objects (-save-temps=obj costs nothing extra);
# tools/micro/audit_{gemm,attn}_asm.sh check it for compiler-made AGPR / scratch / call instructions inside the hand-scheduled kernels
for f in elementwise gemm attention capi; do
  extra=""
  [ "$f" = attention ] && extra="-fno-slp-vectorize -save-temps=obj"
  [ "$f" = gemm ] && extra="-save-temps=obj"
  "$HIPCC" $FLAGS $extra -c "$HERE/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
for f in gemm attention; do   # (best effort: the audit falls back to compiling if the file is not there)
  [ -f "$OUT/$f-hip-amdgcn-amd-amdhsa-gfx950.s" ] && mv -f "$OUT/$f-hip-amdgcn-amd-amdhsa-gfx950.s" "$OUT/$f.gfx950.s" || rm -f "$OUT/$f.gfx950.s"
  rm -f "$OUT/$f"-hip-amdgcn-amd-amdhsa-gfx950.* "$OUT/$f"-host-x86_64-unknown-linux-gnu.* "$OUT/$f".hip-hip-amdgcn-amd-amdhsa.hipfb
done
# the hand-scheduled kernels' register audits run on exactly this build's assembly; a failing audit fails the build
if [ -z "$FDM_SKIP_AUDIT" ]; then
  FDM_AUDIT_ASM="$OUT/gemm.gfx950.s" bash "$HERE/../../tools/micro/audit_gemm_asm.sh" > "$OUT/audit_gemm.log" 2>&1 || { cat "$OUT/audit_gemm.log"; echo "audit_gemm_asm.sh FAILED (FDM_SKIP_AUDIT=1 builds anyway)"; exit 1; }
  FDM_AUDIT_ASM="$OUT/attention.gfx950.s" bash "$HERE/../../tools/micro/audit_attn_asm.sh" > "$OUT/audit_attn.log" 2>&1 || { cat "$OUT/audit_attn.log"; echo "audit_attn_asm.sh FAILED (FDM_SKIP_AUDIT=1 builds anyway)"; exit 1; }
  MASKED=1 FDM_AUDIT_ASM="$OUT/attention.gfx950.s" bash "$HERE/../../tools/micro/audit_attn_asm.sh" > "$OUT/audit_attn_masked.log" 2>&1 || { cat "$OUT/audit_attn_masked.log"; echo "audit_attn_asm.sh (MASKED) FAILED (FDM_SKIP_AUDIT=1 builds anyway)"; exit 1; }
fi
# link under a temporary name and rename: a process that has the old library mapped (another rank, another job) keeps its inode
# instead of seeing a truncated file
TMP_SO="$OUT/.libfastdm_amd.so.$$"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$TMP_SO" "$OUT"/elementwise.o "$OUT"/gemm.o "$OUT"/attention.o "$OUT"/capi.o
mv -f "$TMP_SO" "$OUT/libfastdm_amd.so"
echo "built $OUT/libfastdm_amd.so"
