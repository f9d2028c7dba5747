set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O
cd $R
GBWT_HIP_TRACE_OPEN=1 timeout 900 python tools/c4_open_trace.py full > $O/c4_open.txt 2> $O/c4_open.err; cat $O/c4_open.txt; grep "\[open\]\|\[load\]\|====" $O/c4_open.err | grep -v "checkpoint counts\|summaries\|before the" | tail -21
timeout 600 python -m pytest tests/test_gpu_gfa.py -m gpu -x -q -k "c4_small or synthetic or fixture or translation" 2>&1 | tail -3
