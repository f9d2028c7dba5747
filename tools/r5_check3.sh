set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_gfa.py -m gpu -x -q -k "c4_full_size" ) > $O/t_c4full.log 2>&1; tail -8 $O/t_c4full.log
( time timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q ) > $O/t_dist.log 2>&1; tail -8 $O/t_dist.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_query" ) > $O/t_query.log 2>&1; tail -4 $O/t_query.log
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --sites 100000 --c4-size tiny > $O/bench2.json 2> $O/bench2.err; cut -c1-2500 $O/bench2.json; tail -5 $O/bench2.err
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4; tail -3 $O/bench.err
python - <<'P'
import json
j=json.load(open('gpurun_out/r5d/bench.json'))
print(j['value'], j['parity_checked_paths'])
for k in ('config4','config4_small'):
    c=j.get(k,{}); print(k, {x:c.get(x) for x in ('value','kernel_ms','open_ms','generator_seconds','save_seconds')}, c.get('walk'), c.get('walk_format'))
print(j['search']['unidirectional'], j['search']['bidirectional'])
P
