R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_ac; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o open -- python3 $R/bench.py --steps 3 --warmup 1 --no-config4 --no-search > $O/bench.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs head -40 | cut -c1-200
