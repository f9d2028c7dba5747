R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_aj; mkdir -p $O
mkdir -p $R/gpurun_out; for i in 1 2 3 4 5 6 7 8; do
GBWT_HIP_COMM_TRACE=1 timeout 700 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k loopback > $O/t$i.log 2>&1; tail -1 $O/t$i.log; grep -q passed $O/t$i.log || { tail -80 $O/t$i.log | cut -c1-3000; break; }
done
