export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u > $out/counters_list.txt
wc -l $out/counters_list.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pg1 -o run -- python3 tools/bench_kernels.py gemm > $out/pmc_gemm1.log 2>&1
python3 tools/pmc_summary.py /tmp/pg1 gemm_kernel > $out/pmc_gemm1.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d /tmp/pg2 -o run -- python3 tools/bench_kernels.py gemm > $out/pmc_gemm2.log 2>&1
python3 tools/pmc_summary.py /tmp/pg2 gemm_kernel > $out/pmc_gemm2.txt
cat $out/pmc_gemm1.txt $out/pmc_gemm2.txt | cut -c1-30,80-200
