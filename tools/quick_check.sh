R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03v}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 10 > $O/bench.json 2> $O/bench.err; grep "\[open\]" $O/bench.err | tail -11
python -c "
import json
j=json.load(open('$O/bench.json')); print(j['value'], j['value_cold'], j['open_ms'], j['open']['parse_ms'], j['open']['upload_ms'], j['open']['sample_ms'], j['first_pass_ms'])"
