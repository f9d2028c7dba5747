R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03k}; mkdir -p $O
cd $R
for how in 0 1 2; do
echo "GBWT_HIP_UPLOAD=$how"
GBWT_HIP_UPLOAD=$how GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>&1 >/dev/null | grep "record bytes" | tail -1
done
