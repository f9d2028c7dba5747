set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout 900 python tools/query_call_sweep.py > $O/query_call_sweep.txt 2>&1; cat $O/query_call_sweep.txt | tail -40
