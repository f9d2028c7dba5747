set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
timeout 300 tools/microbench_alloc 1 4 16 48 16 1 > $O/alloc.txt 2>&1; cat $O/alloc.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gfa.py -m gpu -x -q -k "open_flags or line_cache" ) > $O/t.log 2>&1; tail -5 $O/t.log
