export FDM_NO_REBUILD=1
bash tools/gpu_evidence.sh r03a > gpurun_out/r03a_evidence.log 2>&1
bash tools/gpu_evidence_mfma.sh r03a >> gpurun_out/r03a_evidence.log 2>&1
python3 bench.py --model qwen --steps 25 --warmup 0 --no-cpu-baseline --cache-config examples/xcaching/configs/teacache_qwenimage.json --cache-threshold 1.17 --cache-negative > gpurun_out/r03a_teacache_qwen_neg.json 2> gpurun_out/r03a_teacache_qwen_neg.err
bash tools/gpu_evidence_models.sh r03a >> gpurun_out/r03a_evidence.log 2>&1
tail -40 gpurun_out/r03a_evidence.log
tail -c 600 gpurun_out/r03a_teacache_qwen_neg.json
