set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_gfa.py -m gpu -x -q ) > $O/t_gfa.log 2>&1; tail -8 $O/t_gfa.log
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --sites 100000 --c4-size tiny > $O/bench2.json 2> $O/bench2.err; cut -c1-300 $O/bench2.json; grep -n "Error" -B2 -A3 $O/bench2.err | head -30
python - <<'P'
import json
j=json.load(open('gpurun_out/r5e/bench2.json'))
print(json.dumps({k:j.get(k) for k in ('value','shard','other_cut','value_incl_gather')},indent=1))
print(json.dumps(j.get('config4'),indent=1)[:2500])
print(j['config']['final_gather'])
P
timeout 900 python tools/c4_bench.py --size small --passes 5 > $O/c4_small.json 2> $O/c4_small.err; python -c "
import json; j=json.load(open('$O/c4_small.json')); print(j['walk'], j['walk_format'])"
