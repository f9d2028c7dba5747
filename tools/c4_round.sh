# Config 4 on one GPU: tests, the c4 bench object, rocprofv3 kernel stats of the same command.   usage: c4_round.sh TAG
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_c4}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python tools/c4_bench.py --out /dev/shm/c4.gfa > $O/c4.json 2> $O/c4.err; cat $O/c4.json; tail -3 $O/c4.err; rm -f /dev/shm/c4.gfa
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/c4_bench.py --passes 3 > $O/stats.log 2>&1
cd $R
find $O/stats -name "*kernel_stats.csv" -exec head -25 {} \;
find $O -name "*kernel_trace.csv" -size +20M -delete
