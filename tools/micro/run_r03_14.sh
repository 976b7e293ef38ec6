mkdir -p gpurun_out/r03
export FDM_NO_REBUILD=1
timeout 900 python -m pytest tests/test_wan_gpu.py -q -m gpu -k "full_depth_forward" 2>&1 | tail -3
python bench.py --steps 25 --warmup 0 --no-cpu-baseline --cache-config examples/xcaching/configs/teacache_flux.json --cache-threshold 0.886 > gpurun_out/r03/bench_teacache_flux_thr.json 2> gpurun_out/r03/bench_teacache_flux_thr.err
python bench.py --model qwen --steps 25 --warmup 0 --no-cpu-baseline --cache-config examples/xcaching/configs/teacache_qwenimage.json --cache-threshold 1.17 > gpurun_out/r03/bench_teacache_qwen_thr.json 2> gpurun_out/r03/bench_teacache_qwen_thr.err
python - <<'PY'
import json
for n in ("teacache_flux_thr","teacache_qwen_thr"):
    try:
        d=json.loads(open(f"gpurun_out/r03/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "ms/step", round(d["ms_per_step"],2), "s/image", round(d["sec_per_image"],3), "|", d["config"]["workload"][-110:])
    except Exception as e: print(n,"ERR",e)
PY
cat gpurun_out/wan_a14b_full_forward.txt
