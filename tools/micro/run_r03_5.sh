mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03/tests_ops.log
cat gpurun_out/r03/tests_ops.log
