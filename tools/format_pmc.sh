#!/bin/bash
# Counters of the two instantiations of k_format_chunks (tools/format_pmc_driver.py), one group per run; sums over the dispatches of each.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-format_pmc}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
         "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
i=$((i+1))
timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/pass$i -- python3 $R/tools/format_pmc_driver.py > $O/pass$i.log 2>&1
echo "## pass $i: $G"; python3 $R/tools/pmc_sum.py $O/pass$i k_format_chunks; tail -1 $O/pass$i.log
done
find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*counter_collection.csv" -size +5M -delete
