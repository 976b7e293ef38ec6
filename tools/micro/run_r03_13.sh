mkdir -p gpurun_out/r03
export FDM_NO_REBUILD=1
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_qwen_gpu.py tests/test_wan_gpu.py tests/test_flux_gpu.py -q -m gpu -k "maximum_shape or full_depth_forward or radial_mask_and_sparse or c2_bf16 or teacache_at_size" 2>&1 | tail -8 > gpurun_out/r03/tests_fix.log
cat gpurun_out/r03/tests_fix.log
python bench.py --steps 25 --warmup 0 --no-cpu-baseline --cache-config examples/xcaching/configs/teacache_flux.json > gpurun_out/r03/bench_teacache_flux.json 2> gpurun_out/r03/bench_teacache_flux.err
python bench.py --steps 25 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_nocache_flux_25.json 2> /dev/null
python bench.py --model qwen --steps 25 --warmup 0 --no-cpu-baseline --cache-config examples/xcaching/configs/teacache_qwenimage.json > gpurun_out/r03/bench_teacache_qwen.json 2> gpurun_out/r03/bench_teacache_qwen.err
python bench.py --model qwen --steps 25 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_nocache_qwen_25.json 2> /dev/null
python - <<'PY'
import json
for n in ("teacache_flux","nocache_flux_25","teacache_qwen","nocache_qwen_25"):
    try:
        d=json.loads(open(f"gpurun_out/r03/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "ms/step", round(d["ms_per_step"],2), "s/image", round(d["sec_per_image"],3), "|", d["config"]["workload"][-80:])
    except Exception as e: print(n,"ERR",e)
PY
