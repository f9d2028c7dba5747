set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02a}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
cd $R
python3 tools/hbm_traffic.py $O/fetch $O/write k_walk_direct "headline C5k (333334 sites x 5000 haplotypes, mosaic, seed 42)" 13333360000 > $O/hbm_traffic.json; cat $O/hbm_traffic.json | head -30
find $O/stats -name "*kernel_stats.csv" | head; find $O/stats -name "*kernel_stats.csv" -exec head -12 {} \;
