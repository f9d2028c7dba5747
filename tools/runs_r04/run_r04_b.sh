set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_gfa.py tests/test_gpu_parity.py::test_corrupt_large_files_through_the_threaded_open tests/test_gpu_parity.py::test_corrupt_files_never_take_the_device_down -m gpu -x -q --durations=12 > $O/tests.log 2>&1; tail -30 $O/tests.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -5 $O/bench.err
