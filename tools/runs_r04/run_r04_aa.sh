R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_aa; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-config4 --no-search 2>&1 | grep -v amdgpu | grep "checkpoint\|chase\|open_ms\|temporaries" | cut -c1-400
