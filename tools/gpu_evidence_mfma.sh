#!/bin/bash
# MFMA / LDS utilisation counters of the default workload (one PMC pass per counter group, kernel trace only, as the
# profiling guide prescribes).  usage: tools/gpu_evidence_mfma.sh <tag>   (outputs under gpurun_out/<tag>_pmc_*.txt)
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${tag}_$name -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-device-state > /dev/null 2> "$out/${tag}_pmc_$name.err"
  python3 tools/pmc_summary.py /tmp/pmc_${tag}_$name fdm > "$out/${tag}_pmc_$name.txt"
  grep -E "attn_fwd|gemm_kernel<0" "$out/${tag}_pmc_$name.txt"
done
