cd tools/micro
mkdir -p ../../gpurun_out/r03
(echo "== w4"; GEMM_DATA=gauss ./gemm_bench_ab;  echo "== old"; GEMM_DATA=gauss GEMM_OLD=1 ./gemm_bench_ab n; echo "== w4"; GEMM_DATA=gauss ./gemm_bench_ab n;  echo "== old"; GEMM_DATA=gauss GEMM_OLD=1 ./gemm_bench_ab n; echo "== int8 w4"; GEMM_DT=int8 ./gemm_bench_ab; echo "== w4 stamps"; GEMM_DATA=gauss ./gemm_bench_abst n ) 2>&1 | grep -v "vs exact" > ../../gpurun_out/r03/gemm_w4_ab.log
grep -E "==|TOTAL|old vs" ../../gpurun_out/r03/gemm_w4_ab.log | head -40; grep -A22 "== w4 stamps" ../../gpurun_out/r03/gemm_w4_ab.log | grep -E "us|block    0"
