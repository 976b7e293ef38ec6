cd tools/micro
(echo "== persistent"; GEMM_DATA=gauss ./gemm_bench_ab | grep -E "old vs|us |TOTAL"; echo "== one tile per block"; FDM_GEMM_W4_PERSIST=0 GEMM_DATA=gauss ./gemm_bench_ab n | grep -E "us |TOTAL"; echo "== old"; GEMM_OLD=1 GEMM_DATA=gauss ./gemm_bench_ab n | grep -E "us |TOTAL") 2>&1 | grep -v "vs exact"
