R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_g; mkdir -p $O
cd $R
run() { env "$@" timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$*', 'walk kernel', round(j['walk']['kernel_ms'],3), 'stream', round(j['walk']['stream_ms'],3), 'wall', round(j['walk']['wall_ms'],3), 'walk+format', round(j['walk_format']['ms'],3), 'open', round(j['open_ms'],1))"; }
run GBWT_HIP_WALKER_ORDER=1
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_SAMPLE_INTERVAL=512
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_SAMPLE_INTERVAL=1024
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_SAMPLE_INTERVAL=2048
run GBWT_HIP_WALKER_ORDER=0 GBWT_HIP_SAMPLE_INTERVAL=1024
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_DEBUG_DRY_ROWS=1
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_UNIFORM_LOOP=0
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_CATCH_UP=0
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_CHAINS=0
run GBWT_HIP_WALKER_ORDER=1 GBWT_HIP_CHECKPOINT_GAP=128
