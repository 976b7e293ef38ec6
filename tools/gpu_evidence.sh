#!/bin/bash
# Round evidence on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC HBM traffic passes.
# usage: tools/gpu_evidence.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -u
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
tail -c 3000 "$out/${tag}_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-device-state > "$out/${tag}_prof_bench.json" 2> "$out/${tag}_prof.err"
ks=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$ks" "$out/${tag}_kernel_stats.csv"
python3 tools/prof_summary.py "$ks" 6 > "$out/${tag}_summary.txt"
cat "$out/${tag}_summary.txt"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${tag}_$ctr -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-device-state > /dev/null 2> "$out/${tag}_pmc_$ctr.err"
  python3 tools/pmc_summary.py /tmp/pmc_${tag}_$ctr fdm > "$out/${tag}_pmc_$ctr.txt"
  cat "$out/${tag}_pmc_$ctr.txt"
done
