R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_ak; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline_full_size" > $O/tests.log 2>&1; tail -15 $O/tests.log | cut -c1-400
