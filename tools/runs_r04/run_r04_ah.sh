R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python tools/c4_shard_probe.py 2>&1 | grep -v amdgpu | tail -20
