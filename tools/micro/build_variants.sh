#!/bin/bash
# Build gemm_bench variants: build_variants.sh name=flags ...  ->  tools/micro/_ab/gemm_bench_<name> (+ build_<name>.log).
# Everything an A/B builds lives under tools/micro/_ab/ (git-ignored; `rm -rf tools/micro/_ab` before the round ends so it stays
# off the push); NOTHING here writes into fastdm_amd/_lib -- the product library is built by fastdm_amd/csrc/build.sh only, which
# refuses extra flags there.  A whole-library variant: FDM_OUT=tools/micro/_ab/<name> fastdm_amd/csrc/build.sh -D...,
# then FDM_LIB_DIR=tools/micro/_ab/<name> python bench.py --allow-build-flags ...
cd "$(dirname "$0")"
mkdir -p _ab
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result $flags gemm_bench.hip -o _ab/gemm_bench_$name 2> _ab/build_$name.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for spec in "$@"; do name=${spec%%=*}; echo "== $name"; grep -E "error" _ab/build_$name.log | head -3; done
