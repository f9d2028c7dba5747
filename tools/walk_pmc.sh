#!/bin/bash
# Counters of k_walk_direct on config 4's ragged batch (tools/walk_pmc_driver.py), one group per run; sums over its dispatches.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-walk_pmc}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
i=$((i+1))
timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/pass$i -- python3 $R/tools/walk_pmc_driver.py ${2:-small} > $O/pass$i.log 2>&1
echo "## pass $i: $G"; python3 $R/tools/pmc_sum.py $O/pass$i k_walk_direct; grep "LF-steps" $O/pass$i.log
done
find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*counter_collection.csv" -size +5M -delete
