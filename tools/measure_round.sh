# One gpurun call that produces everything a round quotes: parity suite, bench line, rocprofv3 kernel stats of the bench, and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the dominant kernel.   usage: measure_round.sh TAG
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_corrupt_files_never_take_the_device_down > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>&1 >/dev/null | grep "\[open\]" | tail -11 > $O/open_trace.txt; cat $O/open_trace.txt
timeout 600 python tools/open_bench.py --sites 666667 --haplotypes 90 --modes checkpoint,serial 2>&1 | grep -v amdgpu > $O/open_c4.txt; cut -c1-330 $O/open_c4.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-extras > $O/write.log 2>&1
cd $R
python3 tools/hbm_traffic.py $O/fetch $O/write k_walk_direct "headline C5k (333334 sites x 5000 haplotypes, mosaic, seed 42)" 13333360000 > $O/hbm_traffic.json; cat $O/hbm_traffic.json | head -30
find $O/stats -name "*kernel_stats.csv" | head; find $O/stats -name "*kernel_stats.csv" -exec head -30 {} \;
find $O -name "*kernel_trace.csv" -size +20M -delete
