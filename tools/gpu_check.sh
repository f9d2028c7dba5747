# Quick GPU round: the parity suite, the bench line, the 2-rank rehearsal of the N > 1 flow on one GPU (gloo, shared
# device), and last -- it is the one that could upset a box -- the corrupt-file test.   usage: gpu_check.sh TAG
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-check}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --sites 100000 > $O/bench2.json 2> $O/bench2.err; cat $O/bench2.json; tail -5 $O/bench2.err
timeout 900 python -m pytest tests/test_gpu_parity.py::test_corrupt_files_never_take_the_device_down -x -q > $O/mutations.log 2>&1; tail -15 $O/mutations.log
