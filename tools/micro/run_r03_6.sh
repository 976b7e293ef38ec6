mkdir -p gpurun_out/r03
python bench.py --steps 10 --warmup 3 > gpurun_out/r03/bench_w4.json 2> gpurun_out/r03/bench_w4.err
tail -c 3000 gpurun_out/r03/bench_w4.json
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03/tests_all.log
cat gpurun_out/r03/tests_all.log
