#!/bin/bash
mkdir -p gpurun_out/epi4
for rep in 1 2; do
for v in base raggednk; do
  if [ $v = raggednk ]; then export FDM_GEMM_W4_RAGGED_NK=1; else unset FDM_GEMM_W4_RAGGED_NK; fi
  python bench.py --model sd35 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/epi4/sd35_${v}_$rep.json 2>/dev/null
  python bench.py --model sdxl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/epi4/sdxl_${v}_$rep.json 2>/dev/null
done
done
python -m pytest tests/test_ops_gpu.py -x -q -k "matmul or gemm or mm" 2>&1 | tail -3
for f in gpurun_out/epi4/*.json; do python - $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf=d.get('roofline_family') or {}
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],2), 'gemm family ms', round(rf.get('ms_per_step',0),2))
P
done
