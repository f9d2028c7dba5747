R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gfa.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
run() { env "$@" timeout 600 python tools/c4_bench.py --passes 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$*', 'walk kernel', round(j['walk']['kernel_ms'],3), 'stream', round(j['walk']['stream_ms'],3), 'walk+format', round(j['walk_format']['ms'],3), 'fmt stream', round(j['walk_format']['format_stream_ms'],3))"; }
run A=1
run GBWT_HIP_UNIFORM_LOOP=0
run GBWT_HIP_CHAINS=0
run GBWT_HIP_CATCH_UP=0
run GBWT_HIP_SAMPLE_INTERVAL=512
run GBWT_HIP_SAMPLE_INTERVAL=1024
run GBWT_HIP_HELPER_NAPS=2
run GBWT_HIP_LOOKAHEAD_HOPS=31
