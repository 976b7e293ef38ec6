#!/bin/bash
# Build libfastdm_amd.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $*"
pids=()
# attention.hip: the hand-placed MFMA gaps of attn_fwd64_kernel must not be re-packed by the SLP vectoriser (it merges the
# two q-blocks' row-sum chains into v_pk_add_f32 and sinks them below the tile)
for f in elementwise gemm attention capi; do
  extra=""
  [ "$f" = attention ] && extra="-fno-slp-vectorize"
  "$HIPCC" $FLAGS $extra -c "$HERE/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
# link under a temporary name and rename: a process that has the old library mapped (another rank, another job) keeps its inode
# instead of seeing a truncated file
TMP_SO="$OUT/.libfastdm_amd.so.$$"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$TMP_SO" "$OUT"/elementwise.o "$OUT"/gemm.o "$OUT"/attention.o "$OUT"/capi.o
mv -f "$TMP_SO" "$OUT/libfastdm_amd.so"
echo "built $OUT/libfastdm_amd.so"
