#!/bin/bash
# A/B on one box: gemm_w4_kernel also for launches whose M is not a multiple of 256 (FDM_GEMM_W4_RAGGED_M)
mkdir -p gpurun_out/ragged
for rep in 1 2; do
for v in base ragged; do
  if [ $v = ragged ]; then export FDM_GEMM_W4_RAGGED_M=1; else unset FDM_GEMM_W4_RAGGED_M; fi
  python bench.py --model qwen --batch-cfg --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ragged/qwen_bcfg_${v}_$rep.json 2> gpurun_out/ragged/qwen_bcfg_${v}_$rep.err
  python bench.py --model qwen --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ragged/qwen_${v}_$rep.json 2>> gpurun_out/ragged/qwen_bcfg_${v}_$rep.err
  python bench.py --model sdxl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ragged/sdxl_${v}_$rep.json 2>> gpurun_out/ragged/qwen_bcfg_${v}_$rep.err
done
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/ragged/*.json
