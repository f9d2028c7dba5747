R=$GRAFT_REPO_ROOT; cd $R
for st in 4 2 1; do GBWT_HIP_SAMPLE_STRIDE=$st timeout 600 python tools/slab_probe.py 2>&1 | grep -v amdgpu; done
timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu
