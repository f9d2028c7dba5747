R=$GRAFT_REPO_ROOT; cd $R
GBWT_HIP_LIB=$R/tools/probe_csrc/libgbwt_hip.so timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | grep "workgroup\|metric" | cut -c1-400 | tail -12
