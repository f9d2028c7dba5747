R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_ag; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > $O/tests.log 2>&1; tail -30 $O/tests.log | cut -c1-600
