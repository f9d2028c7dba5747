R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_an; mkdir -p $O
python __graft_entry__.py smoke 2>&1 | tail -2
export BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench2.log 2>&1; tail -1 $O/bench2.log | cut -c1-700
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --steps 5 --warmup 2 --shard paths > $O/bench2p.log 2>&1; tail -1 $O/bench2p.log | cut -c1-400
