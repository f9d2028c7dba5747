set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_d; mkdir -p $O
cd $R
timeout 600 python tools/gfa_bench.py --sites 20000 --haplotypes 5000 2>&1 | grep -v amdgpu.ids | tee $O/gfa_bench.txt
for T in 4 1; do
GBWT_HIP_UPLOAD_THREADS=$T GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>&1 >/dev/null | grep "\[open\]\|\[load\]" | tail -28 > $O/open_trace_$T.txt; cat $O/open_trace_$T.txt
GBWT_HIP_UPLOAD_THREADS=$T timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('threads $T', j['value'], j['value_cold'], j['open_ms'], j['open']['parse_ms'], j['open']['upload_ms'], j['open']['sample_ms'], j['first_pass_ms'])"
done
BENCH_ARGS="--no-extras" bash tools/pmc_passes.sh $O/pmc "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum" > $O/pmc_headline.txt 2>&1; cat $O/pmc_headline.txt
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
