#!/bin/bash
# build gemm_bench variants: name=flags ...
cd "$(dirname "$0")"
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result $flags gemm_bench.hip -o gemm_bench_$name 2> build_$name.log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for spec in "$@"; do name=${spec%%=*}; echo "== $name"; grep -E "error" build_$name.log | head -3; done
