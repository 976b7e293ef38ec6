#!/bin/bash
# new auto rule (w4 for every whole-tile problem); ragged-M problems in the 4-wave kernel as an A/B (FDM_GEMM_W4_RAGGED_M)
mkdir -p gpurun_out/epi2
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/epi2/flux_auto.json 2>/dev/null
python bench.py --quant bf16 --height 1024 --width 1024 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/epi2/flux_bf16_auto.json 2>/dev/null
for v in base ragged; do
  if [ $v = ragged ]; then export FDM_GEMM_W4_RAGGED_M=1; else unset FDM_GEMM_W4_RAGGED_M; fi
  python bench.py --model qwen --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/epi2/qwen_$v.json 2>/dev/null
  python bench.py --model qwen --batch-cfg --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/epi2/qwen_bcfg_$v.json 2>/dev/null
  python bench.py --model sdxl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/epi2/sdxl_$v.json 2>/dev/null
  python bench.py --model sd35 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/epi2/sd35_$v.json 2>/dev/null
done
for f in gpurun_out/epi2/*.json; do python - $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=d.get('roofline_family') or {}
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],2), 'gemm family ms', round(rf.get('ms_per_step',0),2))
P
done
