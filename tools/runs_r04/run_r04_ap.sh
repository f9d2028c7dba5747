R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ap; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE"; do
i=$((i+1))
timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/pass$i -- python3 $R/tools/configs.py config4 > $O/pass$i.log 2>&1
echo "## pass $i: $G"; python3 $R/tools/pmc_sum.py $O/pass$i k_format_chunks k_chunk_stats
done
find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*counter_collection.csv" -size +5M -delete
