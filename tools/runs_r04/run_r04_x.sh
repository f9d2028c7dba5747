R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_x; mkdir -p $O
for C in "default 1" "1024 2" "1024 1" "512 4" "512 2" "512 1"; do
set -- $C
if [ $1 = default ]; then unset GBWT_HIP_SAMPLE_INTERVAL; else export GBWT_HIP_SAMPLE_INTERVAL=$1; fi
export GBWT_HIP_SAMPLE_STRIDE=$2
echo "== interval $1 stride $2"
timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
done
