cd tools/micro
mkdir -p ../../gpurun_out/r03
S="2048 12288 3072 0  3328 12288 3072 0  4352 12288 3072 0  5376 12288 3072 0  8704 12288 3072 0"
(echo "== w4"; ./gemm_bench_abst c $S | grep -v "old vs" | grep -v "vs exact"; echo "== old"; GEMM_OLD=1 ./gemm_bench_abst c $S | grep -v "old vs"| grep -v "vs exact") > ../../gpurun_out/r03/gemm_w4_active_cus.log 2>&1
grep -E "==|^M|block    0" ../../gpurun_out/r03/gemm_w4_active_cus.log
