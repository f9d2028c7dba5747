set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e; mkdir -p $O
cd $R
for T in 4 1 4 1; do
GBWT_HIP_UPLOAD_THREADS=$T timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('threads $T', j['value'], j['value_cold'], j['open_ms'], j['open']['parse_ms'], j['open']['upload_ms'], j['open']['sample_ms'], j['first_pass_ms'], j['open']['runtime_init_ms'])"
done
GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>&1 >/dev/null | grep "\[open\]\|\[load\]" | tail -28 > $O/open_trace.txt; cat $O/open_trace.txt

