mkdir -p gpurun_out/r03
export FDM_NO_REBUILD=1
timeout 900 python -m pytest tests/test_flux_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "depth_curve or negative_logits or sparse_attention" 2>&1 | tail -30 > gpurun_out/r03/tests_depth.log
cat gpurun_out/r03/tests_depth.log
cp gpurun_out/c3_depth_parity_fp8.txt gpurun_out/r03/c3_depth_parity_fp8_shipped.txt; cp gpurun_out/c3_depth_parity_bf16.txt gpurun_out/r03/c3_depth_parity_bf16_shipped.txt
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03/bench_shipped_q.json 2>/dev/null
bash fastdm_amd/csrc/build.sh -DFDM_ATT_EXACT_Q=1 > gpurun_out/r03/build_exactq.log 2>&1
timeout 900 python -m pytest tests/test_flux_gpu.py -x -q -m gpu -k "depth_curve" 2>&1 | tail -30 > gpurun_out/r03/tests_depth_exactq.log
cp gpurun_out/c3_depth_parity_fp8.txt gpurun_out/r03/c3_depth_parity_fp8_exactq.txt; cp gpurun_out/c3_depth_parity_bf16.txt gpurun_out/r03/c3_depth_parity_bf16_exactq.txt
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03/bench_exact_q.json 2>/dev/null
bash fastdm_amd/csrc/build.sh > gpurun_out/r03/build_new.log 2>&1
cp gpurun_out/r03/c3_depth_parity_fp8_shipped.txt gpurun_out/c3_depth_parity_fp8.txt; cp gpurun_out/r03/c3_depth_parity_bf16_shipped.txt gpurun_out/c3_depth_parity_bf16.txt
for f in shipped exactq; do echo "== $f"; cat gpurun_out/r03/c3_depth_parity_fp8_$f.txt gpurun_out/r03/c3_depth_parity_bf16_$f.txt; done
python - <<'PY'
import json
for n in ("shipped_q","exact_q"):
    d=json.loads(open(f"gpurun_out/r03/bench_{n}.json").read().strip().splitlines()[-1]); k=d["kernels"]["attn_fwd_bf16"]
    print(n, round(d["ms_per_step"],2), "attention avg us", round(k["avg_ms"]*1e3,1))
PY
