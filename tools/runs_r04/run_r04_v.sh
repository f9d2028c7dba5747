R=$GRAFT_REPO_ROOT; cd $R
for I in default 1024 512; do
if [ $I = default ]; then unset GBWT_HIP_SAMPLE_INTERVAL; else export GBWT_HIP_SAMPLE_INTERVAL=$I; fi
timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
done
