set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O
cd $R
timeout 1500 python tools/c4_interval_sweep.py full > $O/c4_interval_sweep.txt 2> $O/sweep.err; cat $O/c4_interval_sweep.txt; tail -3 $O/sweep.err
