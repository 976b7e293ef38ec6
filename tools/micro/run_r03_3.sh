cd tools/micro
(GEMM_DATA=gauss ./gemm_bench_ab | grep -E "old vs|TOTAL"; GEMM_DATA=gauss ./gemm_bench_abst n 2>&1 | grep -v "vs exact" | grep -v "old vs" | grep -A1 -E "dbl qkv|TOTAL")
