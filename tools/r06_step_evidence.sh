#!/bin/bash
# Round 6 evidence of the headline step on the GPU box: bench line, rocprofv3 kernel stats + the PER-DISPATCH kernel trace joined to the
# GEMM launch shapes (tools/gemm_by_shape.py -> <tag>_gemm_by_shape.csv), optional PMC passes.
# usage: tools/r06_step_evidence.sh <tag> [pmc]
set -u
tag=${1:-r06}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
tail -c 2500 "$out/${tag}_bench.json"
python3 tools/gemm_by_shape.py record "$out/${tag}_gemm_seq.json" > "$out/${tag}_gemm_seq.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-device-state > "$out/${tag}_prof_bench.json" 2> "$out/${tag}_prof.err"
ks=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
kt=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
cp "$ks" "$out/${tag}_kernel_stats.csv"
python3 tools/prof_summary.py "$ks" 6 > "$out/${tag}_summary.txt"
cat "$out/${tag}_summary.txt"
python3 tools/gemm_by_shape.py join "$kt" "$out/${tag}_gemm_seq.json" "$out/${tag}_gemm_by_shape.csv"
if [ "${2:-}" = pmc ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${tag}_$ctr -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-device-state > /dev/null 2> "$out/${tag}_pmc_$ctr.err"
    python3 tools/pmc_summary.py /tmp/pmc_${tag}_$ctr fdm > "$out/${tag}_pmc_$ctr.txt"
    cat "$out/${tag}_pmc_$ctr.txt"
  done
fi
