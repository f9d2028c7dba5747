R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "indel or chopped or chained or ragged or segmented or random" > $O/tests.log 2>&1; tail -3 $O/tests.log
run() { env "$@" timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$*', 'walk kernel', round(j['walk']['kernel_ms'],3), 'walk+format', round(j['walk_format']['ms'],3))"; }
run A=1
run GBWT_HIP_CATCH_UP=0
run A=2
for A in 1 0; do
GBWT_HIP_CATCH_UP=$A timeout 600 python tools/configs.py secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('catch_up $A secondary', j['kernel_ms'], j['value'])"
done
timeout 600 python tools/indel_bench.py 2>&1 | grep -v amdgpu | tail -8
