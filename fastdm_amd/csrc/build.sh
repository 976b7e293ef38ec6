#!/bin/bash
# Build libfastdm_amd.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $*"
pids=()
for f in elementwise gemm attention capi; do
  "$HIPCC" $FLAGS -c "$HERE/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfastdm_amd.so" "$OUT"/elementwise.o "$OUT"/gemm.o "$OUT"/attention.o "$OUT"/capi.o
echo "built $OUT/libfastdm_amd.so"
