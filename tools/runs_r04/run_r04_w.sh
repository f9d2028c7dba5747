R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented or ragged or headline or c2 or indel or output" > $O/tests.log 2>&1; tail -2 $O/tests.log
for I in default 1024 512; do
if [ $I = default ]; then unset GBWT_HIP_SAMPLE_INTERVAL; else export GBWT_HIP_SAMPLE_INTERVAL=$I; fi
timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
done
