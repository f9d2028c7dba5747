set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_gfa.py -m gpu -x -q > $O/tests.log 2>&1; tail -8 $O/tests.log
timeout 600 python tools/gfa_bench.py --sites 20000 --haplotypes 5000 2>&1 | grep -v amdgpu.ids | tee $O/gfa_bench.txt
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras > $O/bench2.json 2> $O/bench2.err; cat $O/bench2.json | cut -c1-1500; tail -5 $O/bench2.err
