#!/bin/bash
tag=r06s_qwen
out=$PWD/gpurun_out
export TMPDIR=/tmp
python3 tools/gemm_by_shape.py record "$out/${tag}_gemm_seq.json" --model qwen --batch-cfg > "$out/${tag}_gemm_seq.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 bench.py --model qwen --steps 5 --warmup 1 --no-cpu-baseline --no-device-state > "$out/${tag}_prof_bench.json" 2> "$out/${tag}_prof.err"
kt=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/gemm_by_shape.py join "$kt" "$out/${tag}_gemm_seq.json" "$out/${tag}_gemm_by_shape.csv"
