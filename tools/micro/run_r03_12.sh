mkdir -p gpurun_out/r03
timeout 3000 python -m pytest tests/test_ops_gpu.py tests/test_qwen_gpu.py tests/test_wan_gpu.py -q -m gpu -k "maximum_shape or full_depth_forward_at_1328 or radial_mask_and_sparse" 2>&1 | grep -E "^E  |Error|assert|FAILED|passed|failed" | grep -v "where" | head -40 > gpurun_out/r03/tests_fail.log
cat gpurun_out/r03/tests_fail.log
