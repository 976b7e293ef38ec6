cd tools/micro
mkdir -p ../../gpurun_out/r03
# how does the time of ONE round depend on the number of active CUs?  N = 3072 (12 tile columns), K = 12288
./gemm_bench_k64st c 2048 12288 3072 0  3328 12288 3072 0  4352 12288 3072 0  5376 12288 3072 0  6400 12288 3072 0  6656 12288 3072 0  8704 12288 3072 0 > ../../gpurun_out/r03/gemm_active_cus.log 2>&1
cat ../../gpurun_out/r03/gemm_active_cus.log
