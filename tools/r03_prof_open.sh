R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03f}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_corrupt_files_never_take_the_device_down > $O/tests.log 2>&1; tail -5 $O/tests.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/open_stats -- python3 $R/tools/open_bench.py --reps 2 --passes 3 --modes checkpoint > $O/open_stats.log 2>&1
cat $O/open_stats.log | grep -v "^W2\|^E2\|rocprof" | cut -c1-420 | tail -5
find $O/open_stats -name "*kernel_stats.csv" -exec head -40 {} \;
find $O/open_stats -name "*kernel_trace.csv" -delete
