set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "config4" ) > $O/t_dist.log 2>&1; tail -8 $O/t_dist.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "open_flags" ) > $O/t_flags.log 2>&1; tail -8 $O/t_flags.log
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --sites 100000 --c4-size medium > $O/bench2.json 2> $O/bench2.err; grep -n "Error" -B2 -A3 $O/bench2.err | head -30
python - <<'P'
import json
txt=[l for l in open('gpurun_out/r5h/bench2.json') if l.startswith('{')][-1]
j=json.loads(txt); c=j['config4']
print(json.dumps({k:v for k,v in c.items() if k!='ranks'},indent=1)[:2500]); print(c['ranks'][1])
P
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetchcal -- $R/tools/microbench_fetch > $O/fetchcal.log 2>&1; tail -8 $O/fetchcal.log
cd $R; python3 tools/fetch_calibration.py $O/fetchcal > $O/fetch_calibration.txt; cat $O/fetch_calibration.txt
find $O -name "*kernel_trace.csv" -delete
