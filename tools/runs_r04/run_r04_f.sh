set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gfa.py -m gpu -x -q -k "ragged or segmented or c4 or indel or chopped or chained or random or variants or output_paths or c2 or fixture" > $O/tests.log 2>&1; tail -4 $O/tests.log
for W in 1 0 1 0; do
GBWT_HIP_WALKER_ORDER=$W timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('order $W walk', j['walk']['kernel_ms'], j['walk']['stream_ms'], j['walk']['wall_ms'], 'walk+format', j['walk_format']['ms'], j['walk_format']['walk_kernel_ms'], j['walk_format']['format_stream_ms'])"
done
for W in 1 0; do
GBWT_HIP_WALKER_ORDER=$W timeout 600 python tools/configs.py secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('order $W secondary', j['kernel_ms'], j['value'], j['value_kernel'])"
done
GBWT_HIP_WALKER_ORDER=1 timeout 600 python tools/ragged_bench.py 2>&1 | grep -v amdgpu | tail -6
GBWT_HIP_WALKER_ORDER=0 timeout 600 python tools/ragged_bench.py 2>&1 | grep -v amdgpu | tail -6
