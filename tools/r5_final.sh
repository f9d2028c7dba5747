set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5z; mkdir -p $O
cd $R
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
GBWT_HIP_TRACE_GFA=1 timeout 1200 python tools/c4_bench.py --size full --passes 2 --out /dev/shm/gbwt_c4_stated.gfa > $O/c4_full_file.json 2> $O/c4_full_file.err; python -c "
import json; j=json.load(open('$O/c4_full_file.json')); print(j['whole_file'], j['walk_format'])"; grep "\[gfa\]" $O/c4_full_file.err | tail -4; rm -f /dev/shm/gbwt_c4_stated.gfa
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --sites 100000 --c4-size small > $O/bench2.json 2> $O/bench2.err; grep -c "^{" $O/bench2.json; grep -n "Error" -A3 $O/bench2.err | head -20
