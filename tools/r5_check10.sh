set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "open_flags or parts_of_rows or fixture" ) > $O/t_flags.log 2>&1; tail -8 $O/t_flags.log
( time timeout 900 python -m pytest tests/test_gpu_gfa.py tests/test_gpu_dist.py -m gpu -x -q ) > $O/t_gfa.log 2>&1; tail -5 $O/t_gfa.log
timeout 900 python tools/c4_bench.py --size full --passes 3 > $O/c4_full.json 2> $O/c4_full.err; python -c "
import json; j=json.load(open('$O/c4_full.json')); print(j['open_ms'], j['walk'], j['walk_format'], j['memory'])"; tail -3 $O/c4_full.err
