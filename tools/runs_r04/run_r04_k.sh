R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_k; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for A in 1 0 1 0; do
GBWT_HIP_ALL4=$A timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('all4 $A headline', j['value'], j['roofline']['kernel_ms'])"
done
for A in 1 0; do
GBWT_HIP_ALL4=$A timeout 600 python tools/configs.py secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('all4 $A secondary', j['kernel_ms'], j['value'])"
done
