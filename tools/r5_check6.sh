set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/tests.log 2>&1; tail -6 $O/tests.log
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4; tail -3 $O/bench.err
python - <<'P'
import json
j=json.load(open('gpurun_out/r5g/bench.json'))
print(j['value'], j['parity_checked_paths'], j['ms_per_step'])
for k in ('secondary','high_degree','search','config4','config4_small'):
    c=j.get(k,{}); print(k, c.get('value'), c.get('kernel_ms'), c.get('cpu_baseline'))
print(j['config4'].get('walk_format'))
print(j['config4_small'].get('walk_format'), j['config4_small'].get('whole_file'))
P
