set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean_extract or walk_loop_variants or segmented_extraction or walk_tables or open_flags or indel or chained or parts_of_rows" ) > $O/t.log 2>&1; tail -6 $O/t.log
( time timeout 900 python -m pytest tests/test_gpu_gfa.py tests/test_gpu_dist.py -m gpu -x -q ) > $O/t2.log 2>&1; tail -4 $O/t2.log
# A/B: catch-up on the one-step descriptors (1) against the two-step ones (2), secondary + headline, full handles
for C in 1 2 1 2; do GBWT_HIP_CATCH_UP=$C timeout 600 python tools/configs.py secondary 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('catch_up $C secondary', round(j['kernel_ms'],3), 'ms', round(j['value']/1e9,1), 'G')"; done
timeout 900 python tools/c4_bench.py --size full --passes 3 > $O/c4_full.json 2> $O/c4_full.err; python -c "
import json; j=json.load(open('$O/c4_full.json')); print(j['open_ms'], j['walk'], j['walk_format']['ms'], j['memory'])"; tail -2 $O/c4_full.err
