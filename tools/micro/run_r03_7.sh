mkdir -p gpurun_out/r03
export FDM_NO_REBUILD=1
python bench.py --steps 10 --warmup 3 > gpurun_out/r03/bench_ab_w4.json 2> gpurun_out/r03/bench_ab_w4.err
bash fastdm_amd/csrc/build.sh -DFDM_GEMM_W4=0 > gpurun_out/r03/build_old.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/r03/bench_ab_old.json 2> gpurun_out/r03/bench_ab_old.err
bash fastdm_amd/csrc/build.sh > gpurun_out/r03/build_new.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/r03/bench_ab_w4b.json 2> gpurun_out/r03/bench_ab_w4b.err
python - <<'PY'
import json
for n in ("w4","old","w4b"):
    try:
        d=json.loads(open(f"gpurun_out/r03/bench_ab_{n}.json").read().strip().splitlines()[-1])
        k=d["kernels"]
        print(n, round(d["ms_per_step"],2), {x:(round(k[x]["avg_ms"]*1e3,1), round(k[x]["ms_per_step"],2)) for x in k})
    except Exception as e: print(n, "ERR", e)
PY
