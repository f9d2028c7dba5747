R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_h; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gfa.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
run() { env "$@" timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$*', 'walk kernel', round(j['walk']['kernel_ms'],3), 'stream', round(j['walk']['stream_ms'],3), 'wall', round(j['walk']['wall_ms'],3), 'walk+format', round(j['walk_format']['ms'],3), 'open', round(j['open_ms'],1))"; }
run GBWT_HIP_ALIGN_SEGMENTS=1
run GBWT_HIP_ALIGN_SEGMENTS=0
run GBWT_HIP_ALIGN_SEGMENTS=1 GBWT_HIP_SAMPLE_INTERVAL=1024
run GBWT_HIP_ALIGN_SEGMENTS=0 GBWT_HIP_SAMPLE_INTERVAL=1024
for A in 1 0 1 0; do
GBWT_HIP_ALIGN_SEGMENTS=$A timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('align $A headline', j['value'], j['roofline']['kernel_ms'])"
done
for A in 1 0; do
GBWT_HIP_ALIGN_SEGMENTS=$A timeout 600 python tools/configs.py secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('align $A secondary', j['kernel_ms'], j['value'])"
GBWT_HIP_ALIGN_SEGMENTS=$A timeout 600 python tools/configs.py high_degree 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('align $A high_degree', j['kernel_ms'], j['value_kernel'])"
done
