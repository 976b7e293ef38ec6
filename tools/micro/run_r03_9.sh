mkdir -p gpurun_out/r03
export FDM_NO_REBUILD=1
run() { # name, env, args
  local n=$1; shift; local e=$1; shift
  env FDM_GEMM_KERNEL=$e python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/ab_$n.json 2> gpurun_out/r03/ab_$n.err
}
run flux_auto auto --steps 10 --warmup 3
run flux_w8 w8 --steps 10 --warmup 3
run flux_w4 w4 --steps 10 --warmup 3
run qwen_w4 w4 --model qwen --steps 6 --warmup 2
run qwen_w8 w8 --model qwen --steps 6 --warmup 2
run fluxbf16_w4 w4 --quant bf16 --height 1024 --width 1024 --steps 8 --warmup 2
run fluxbf16_w8 w8 --quant bf16 --height 1024 --width 1024 --steps 8 --warmup 2
run fluxint8_w4 w4 --quant int8 --steps 8 --warmup 2
run fluxint8_w8 w8 --quant int8 --steps 8 --warmup 2
run sd35_auto auto --model sd35 --steps 8 --warmup 2
run sd35_w8 w8 --model sd35 --steps 8 --warmup 2
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/ab_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get("kernels",{})
        print(f.split("ab_")[1][:-5].ljust(14), round(d["ms_per_step"],2), {x:(round(k[x]["avg_ms"]*1e3,1), round(k[x]["ms_per_step"],2)) for x in k if 'gemm' in x})
    except Exception as e: print(f, "ERR", e)
PY
