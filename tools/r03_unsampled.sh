R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03x}; mkdir -p $O
cd $R
{
for env in "GBWT_HIP_SAMPLE_INTERVAL=0" "GBWT_HIP_SAMPLE_INTERVAL=0 GBWT_HIP_CATCH_UP=0" "GBWT_HIP_SAMPLE_INTERVAL=0 GBWT_HIP_UNIFORM_LOOP=0" "GBWT_HIP_SAMPLE_INTERVAL=0 GBWT_HIP_PACKED_BLOCKS=0" "GBWT_HIP_SAMPLE_INTERVAL=0 GBWT_HIP_VMM=0" "GBWT_HIP_SAMPLE_INTERVAL=0 GBWT_HIP_RING_SLOTS=64"; do
echo "## $env"; env $env timeout 600 python tools/sweep.py --sites 333334 --configs 0:0:16 --reps 4 2>&1 | grep -v amdgpu
done
echo "## host buffer"; for t in 4 8 16; do GBWT_HIP_COPY_THREADS=$t timeout 600 python tools/host_buffer_bench.py 2>&1 | grep -v amdgpu | tail -3; done
} > $O/unsampled.txt 2>&1; cat $O/unsampled.txt
