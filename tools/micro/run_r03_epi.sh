#!/bin/bash
# A/B of the GEMM epilogue rewrite (operands requested up front, whole-tile store fast path): old = previous revision of gemm.hip
cd tools/micro
for rep in 1 2; do
  for v in old new; do
    echo "== $v w8 (rep $rep)"; FDM_GEMM_KERNEL=w8 ./gemm_bench_$v n 2>&1 | tail -14
    echo "== $v w4 (rep $rep)"; FDM_GEMM_KERNEL=w4 ./gemm_bench_$v n 2>&1 | tail -14
  done
done
echo "== AB correctness (new source: w8 vs w4 elementwise + host reference)"; ./gemm_bench_ab c 512 3072 768 0 768 3072 512 1 1024 12288 512 3 2>&1 | tail -12
