R=$GRAFT_REPO_ROOT; cd $R
P=$R/tools/probe_csrc/libgbwt_hip.so
run() { echo "== $1"; shift; env "$@" timeout 600 python tools/shard_probe.py 2>&1 | grep -v amdgpu | grep "^N=1\|^N=8"; }
run "default build, default knobs"
run "default build, ring 32 / piece 16" GBWT_HIP_RING_SLOTS=32 GBWT_HIP_ROW_PIECE=16
run "96-VGPR build (5 waves/SIMD), ring 32 / piece 16" GBWT_HIP_LIB=$P GBWT_HIP_RING_SLOTS=32 GBWT_HIP_ROW_PIECE=16
run "96-VGPR build, default knobs (LDS keeps 8 workgroups per CU)" GBWT_HIP_LIB=$P
run "default build, default knobs"
