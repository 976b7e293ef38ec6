export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --model sdxl --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-600
python bench.py --model sdxl --steps 10 --warmup 3 --graph 2>&1 | tail -1 | cut -c1-330
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sdxl -o run -- python3 $GRAFT_REPO_ROOT/bench.py --model sdxl --steps 4 --warmup 1 --no-kernel-timing > /dev/null 2>&1
ks=$(find /tmp/prof_sdxl -name "*kernel_stats.csv" | head -1)
cp $ks $GRAFT_REPO_ROOT/gpurun_out/r01g_sdxl_kernel_stats.csv
cd $GRAFT_REPO_ROOT; python3 tools/prof_summary.py $ks 5 | cut -c1-150 | head -40
