#!/bin/bash
# whole-step A/B of the kernel choice after the epilogue rewrite: auto (shipped rule) / w4 everywhere / w8 everywhere
mkdir -p gpurun_out/epi
for rep in 1 2; do
  for v in auto w4 w8; do
    FDM_GEMM_KERNEL=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/epi/flux_${v}_$rep.json 2> gpurun_out/epi/flux_${v}_$rep.err
  done
done
for v in auto w4; do FDM_GEMM_KERNEL=$v python bench.py --quant int8 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/epi/flux_int8_${v}.json 2>/dev/null; done
for f in gpurun_out/epi/*.json; do python - $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],2), {n:round(v['avg_ms']*1e3,1) for n,v in k.items() if 'gemm_w8' in n}, round(k.get('attn_fwd_bf16',{}).get('avg_ms',0)*1e3,1))
P
done
