#!/bin/bash
# build_attn_variants.sh name=flags ...  -> tools/micro/_ab/libattn_<name>.so from fastdm_amd/csrc/attention.hip (+ -D flags), and
# _ab/attn_ab.  Like build_variants.sh: outputs only under tools/micro/_ab/, never fastdm_amd/_lib.
cd "$(dirname "$0")"
mkdir -p _ab
HIPCC=/opt/rocm/bin/hipcc
BASE="-I../../fastdm_amd/csrc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-result"
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  src=../../fastdm_amd/csrc/attention.hip
  case "$flags" in SRC:*) src=${flags#SRC:}; src=${src%% *}; flags=${flags#SRC:$src}; ;; esac
  ( $HIPCC $BASE $flags -shared $src attn_variant_stub.hip -o _ab/libattn_$name.so 2> _ab/build_attn_$name.log || echo "BUILD FAILED: $name" ) &
  pids+=($!)
done
$HIPCC -O2 -std=c++17 attn_ab.cpp -o _ab/attn_ab -ldl 2> _ab/build_attn_ab.log || echo "attn_ab build failed"
for p in "${pids[@]}"; do wait $p; done
ls -la _ab/libattn_*.so _ab/attn_ab 2>/dev/null | awk '{print $5, $9}'
