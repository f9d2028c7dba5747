set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
( time timeout 1500 python tools/c4_bench.py --size full --passes 3 > $O/c4_full.json ) 2> $O/c4_full.err; cut -c1-3000 $O/c4_full.json; tail -25 $O/c4_full.err
