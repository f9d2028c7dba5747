R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for i in 1 2; do timeout 600 python tools/configs.py search 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('search', j['unidirectional']['kernel_ms'], j['bidirectional']['kernel_ms'], j['value'])"; done
timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('c4 walk kernel', round(j['walk']['kernel_ms'],3), 'walk+format', round(j['walk_format']['ms'],3), 'open', round(j['open_ms'],1), j['memory'])"
