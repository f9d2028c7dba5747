R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for L in 1 0 1 0; do
GBWT_HIP_RING_LAYOUT=$L timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('layout $L headline', j['value'], j['roofline']['kernel_ms'])"
done
for L in 1 0; do
GBWT_HIP_RING_LAYOUT=$L timeout 600 python tools/configs.py secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('layout $L secondary', j['kernel_ms'], j['value'])"
GBWT_HIP_RING_LAYOUT=$L timeout 600 python tools/configs.py high_degree 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('layout $L high_degree', j['kernel_ms'], j['value_kernel'])"
GBWT_HIP_RING_LAYOUT=$L timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('layout $L c4 walk', j['walk']['kernel_ms'], 'walk+format', j['walk_format']['ms'])"
done
