set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5o; mkdir -p $O
cd $R
GBWT_HIP_TRACE_OPEN=1 timeout 900 python tools/c4_open_trace.py full > $O/c4_open.txt 2> $O/c4_open.err; cat $O/c4_open.txt; grep "\[open\]\|\[load\]\|====" $O/c4_open.err | grep -v "checkpoint counts\|summaries\|before the" | tail -60
GBWT_HIP_TRACE_OPEN=1 timeout 600 python tools/c4_open_trace.py small > $O/c4_open_small.txt 2> $O/c4_open_small.err; cat $O/c4_open_small.txt
timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline open', j['open'])"
