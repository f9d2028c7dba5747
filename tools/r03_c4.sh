R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03h}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gfa.py tests/test_gpu_parity.py -m gpu -x -q -k "c4 or path_modes or rle or long_runs or segmented" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python tools/gfa_sharded.py --contigs 24 --fragments 3 --haplotypes 40 --sites 200 --check --oracle > $O/c4_small.txt 2>&1; cat $O/c4_small.txt
timeout 900 python tools/gfa_sharded.py --check > $O/c4_default.txt 2>&1; cat $O/c4_default.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/gfa_sharded.py --contigs 24 --fragments 4 --haplotypes 40 --sites 400 --backend gloo --share-gpu --check --oracle > $O/c4_two_ranks.txt 2>&1; tail -4 $O/c4_two_ranks.txt
