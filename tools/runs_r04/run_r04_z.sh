R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_z; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented or ragged or headline or c2 or indel or output" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_bench.py -m gpu -x -q > $O/tests2.log 2>&1; tail -2 $O/tests2.log
unset GBWT_HIP_SAMPLE_INTERVAL GBWT_HIP_SAMPLE_STRIDE
timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
GBWT_HIP_SAMPLE_COARSE=1 timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-config4 --no-search 2>&1 | grep -v amdgpu | tail -60
