#!/bin/bash
# A/B: plain whole-tile epilogue of gemm_w4_kernel in two 64-row halves (stores of half 0 drain under pass 1 of half 1)
cd tools/micro
for rep in 1 2 3; do
  for v in old new; do echo "== $v (rep $rep)"; ./gemm_bench_$v n 2>&1 | tail -8; done
done
echo "== stamps + correctness"; ./gemm_bench_st c 8192 3072 9216 0 8192 3072 12288 1 2>&1 | grep -v "vs exact"
