R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for L in new old new old; do
if [ $L = old ]; then export GBWT_HIP_LIB=$R/tools/probe_csrc/libgbwt_hip.so; else unset GBWT_HIP_LIB; fi
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L headline', j['value'], j['roofline']['kernel_ms'])"
done
for L in new old; do
if [ $L = old ]; then export GBWT_HIP_LIB=$R/tools/probe_csrc/libgbwt_hip.so; else unset GBWT_HIP_LIB; fi
timeout 600 python tools/configs.py secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L secondary', j['kernel_ms'], j['value'])"
timeout 600 python tools/configs.py high_degree 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L high_degree', j['kernel_ms'], j['value_kernel'])"
timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L c4 walk', j['walk']['kernel_ms'], 'walk+format', j['walk_format']['ms'])"
done
