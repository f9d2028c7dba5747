R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gfa.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
GBWT_HIP_TRACE_GFA=1 timeout 600 python tools/c4_bench.py --passes 2 --out /dev/shm/c4.gfa 2>&1 | grep "\[gfa\]\|whole_file" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('[gfa]'): print(l.strip())
    else: print(json.loads(l)['whole_file'])"; rm -f /dev/shm/c4.gfa
