#!/bin/bash
# Qwen-Image evidence: bench line + rocprofv3 kernel stats.  usage: tools/gpu_evidence_qwen.sh <tag>
set -u
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --model qwen --steps 10 --warmup 2 > "$out/${tag}_qwen_bench.json" 2> "$out/${tag}_qwen_bench.err"
tail -c 1500 "$out/${tag}_qwen_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q$tag -o run -- python3 bench.py --model qwen --steps 3 --warmup 1 > /dev/null 2> "$out/${tag}_qwen_prof.err"
ks=$(find /tmp/prof_q$tag -name "*kernel_stats.csv" | head -1)
cp "$ks" "$out/${tag}_qwen_kernel_stats.csv"
python3 tools/prof_summary.py "$ks" 8 > "$out/${tag}_qwen_summary.txt"
cat "$out/${tag}_qwen_summary.txt"
