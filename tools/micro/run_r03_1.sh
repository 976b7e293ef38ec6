cd tools/micro
mkdir -p ../../gpurun_out/r03
for r in 1 2; do for v in k64 k128; do echo "== $v round $r"; ./gemm_bench_$v; done; done > ../../gpurun_out/r03/gemm_k128_ab.log 2>&1
for v in k64st k128st; do echo "== $v"; ./gemm_bench_$v; done > ../../gpurun_out/r03/gemm_k128_stamps.log 2>&1
cd ../..
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/libnames -o lib -- python3 tools/micro/library_kernel_names.py > gpurun_out/r03/libnames.log 2>&1
tail -30 gpurun_out/r03/gemm_k128_ab.log
