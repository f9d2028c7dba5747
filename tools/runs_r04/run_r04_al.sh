R=$GRAFT_REPO_ROOT; cd $R
GBWT_HIP_LIB=$R/tools/probe_csrc/libgbwt_hip.so timeout 600 python tools/slab_probe.py 2>&1 | grep -v amdgpu | grep "N=1 \|N=8 rank 4\|extract\]" | tail -30
