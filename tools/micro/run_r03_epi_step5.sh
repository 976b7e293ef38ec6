#!/bin/bash
# after batching the 4-wave kernel's split-K merge loads: ragged-M launches of < 2 rounds in the 4-wave kernel too? (FDM_GEMM_W4_ANY_M)
mkdir -p gpurun_out/epi5
for rep in 1 2; do
for v in base anym; do
  if [ $v = anym ]; then export FDM_GEMM_W4_ROUNDS=1; else unset FDM_GEMM_W4_ROUNDS; fi
  python bench.py --model qwen --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/epi5/qwen_${v}_$rep.json 2>/dev/null
  python bench.py --model sdxl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/epi5/sdxl_${v}_$rep.json 2>/dev/null
done
done
true
for f in gpurun_out/epi5/*.json; do python - $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=d.get('roofline_family') or {}
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],2), 'gemm family ms', round(rf.get('ms_per_step',0),2))
P
done
