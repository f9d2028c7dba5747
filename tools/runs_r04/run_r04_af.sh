R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_af; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "parts_of_rows or segmented or ragged" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > $O/tests2.log 2>&1; tail -5 $O/tests2.log
timeout 600 python tools/slab_probe.py 2>&1 | grep -v amdgpu
export BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench2.log 2>&1; tail -3 $O/bench2.log | cut -c1-1500
