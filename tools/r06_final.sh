# round 6, the last GPU call: tools/measure_round.sh (tests, bench line, kernel stats, PMC traffic of every config) + the counter sets of
# rounds 4 / 5 re-taken on this build (headline: tools/pmc_passes.sh; config 4's ragged walk: tools/walk_pmc.sh) + __graft_entry__.smoke()
R=$GRAFT_REPO_ROOT; cd $R
bash tools/measure_round.sh r06
O=$R/gpurun_out/r06
BENCH_ARGS="--no-extras" bash tools/pmc_passes.sh $O/pmc_headline \
  "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum" > $O/pmc_headline.txt 2>&1
cd $R; find $O/pmc_headline -name "*.csv" -size +1M -delete
bash tools/walk_pmc.sh r06/c4_walk_pmc small > $O/c4_walk_pmc.txt 2>&1
cd $R; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
tail -30 $O/pmc_headline.txt
