mkdir -p gpurun_out/r03
export FDM_NO_REBUILD=1
run() { local n=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/sp_$n.json 2> gpurun_out/r03/sp_$n.err; }
run flux_a --steps 10 --warmup 3
run qwen_a --model qwen --steps 6 --warmup 2
bash fastdm_amd/csrc/build.sh -DFDM_GEMM_SPREAD_DMA=1 > gpurun_out/r03/build_spread.log 2>&1
run flux_b --steps 10 --warmup 3
run qwen_b --model qwen --steps 6 --warmup 2
bash fastdm_amd/csrc/build.sh > gpurun_out/r03/build_new.log 2>&1
run flux_a2 --steps 10 --warmup 3
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/sp_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get("kernels",{})
        print(f.split("sp_")[1][:-5].ljust(10), round(d["ms_per_step"],2), {x:(round(k[x]["avg_ms"]*1e3,1), round(k[x]["ms_per_step"],2)) for x in k if 'gemm_w8' in x}, d.get("roofline_family",{}).get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
