R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ar; mkdir -p $O; cd $R
P=$R/tools/probe_csrc/libgbwt_hip.so
show() { python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=j['walk_format']; print('$1: walk+format %.3f ms (walk kernel %.3f, format stream %.3f)' % (w['ms'], w['walk_kernel_ms'], w['format_stream_ms']))"; }
python3 tools/configs.py config4 2>/dev/null | show shipped
GBWT_HIP_LIB=$P python3 tools/configs.py config4 2>/dev/null | show "chunk->path table + node ids ahead"
python3 tools/configs.py config4 2>/dev/null | show shipped
GBWT_HIP_LIB=$P python3 tools/configs.py config4 2>/dev/null | show "chunk->path table + node ids ahead"
GBWT_HIP_LIB=$P timeout 900 python -m pytest tests/test_gpu_gfa.py -m gpu -x -q 2>&1 | tail -1
