R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_aq; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_gfa.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-400
for i in 1 2; do python3 tools/configs.py config4 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=j['walk_format']; print('config4: walk+format %.3f ms (walk kernel %.3f, format stream %.3f), text %.2f GB/s' % (w['ms'], w['walk_kernel_ms'], w['format_stream_ms'], w['text_GB_per_s']))"; done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pass2 -- python3 $R/tools/configs.py config4 > $O/pass2.log 2>&1
python3 $R/tools/pmc_sum.py $O/pass2 k_format_chunks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/configs.py config4 > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec head -6 {} \; | cut -d, -f1-4 | cut -c1-160
find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*counter_collection.csv" -size +5M -delete
