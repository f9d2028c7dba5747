R=$GRAFT_REPO_ROOT; cd $R
GBWT_HIP_TRACE_GFA=1 timeout 600 python tools/c4_bench.py --passes 2 --out /dev/shm/c4.gfa 2>&1 | grep "\[gfa\]\|whole_file" | cut -c1-400; rm -f /dev/shm/c4.gfa
