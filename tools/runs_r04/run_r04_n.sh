R=$GRAFT_REPO_ROOT; cd $R
for W in 3 6 10; do
GBWT_HIP_GFA_WRITERS=$W GBWT_HIP_TRACE_GFA=1 timeout 600 python tools/c4_bench.py --passes 2 --out /dev/shm/c4.gfa 2>&1 | grep "\[gfa\]\|whole_file" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('[gfa]'): print('W=$W', l.strip())
    else: print('W=$W', json.loads(l)['whole_file'])"; rm -f /dev/shm/c4.gfa
done
