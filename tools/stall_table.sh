#!/bin/bash
# Where the wave cycles of the step's MFMA kernels go: three rocprofv3 PMC passes over one forward of bench.py
# (kernel trace only, counters in their own runs), then tools/stall_table.py -> gpurun_out/<tag>_stall_table.txt.
# SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall: MFMA RAW, busy pipe) + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES
# (MI355X_MICROARCH.md, rocprofv3 PMC slots); WAVE / WAIT / ACTIVE count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles.
tag=${1:-r06}
shift
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_WAVES"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/stall_${tag}_$i -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-device-state "$@" > /dev/null 2> "$out/${tag}_stall_$i.err"
done
python3 tools/stall_table.py /tmp/stall_${tag}_1 /tmp/stall_${tag}_2 /tmp/stall_${tag}_3 > "$out/${tag}_stall_table.txt"
cat "$out/${tag}_stall_table.txt"
