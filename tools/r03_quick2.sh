R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03y}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline_full_size or two_threads or c3_scale" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/open_bench.py --sites 666667 --haplotypes 90 --modes checkpoint,serial 2>&1 | grep -v amdgpu | cut -c1-330
timeout 600 python tools/open_bench.py --sites 666667 --haplotypes 90 --modes "checkpoint:SAMPLE_INTERVAL=128,checkpoint:SAMPLE_INTERVAL=1024" --reps 1 2>&1 | grep -v amdgpu | cut -c1-330
