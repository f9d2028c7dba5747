R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03i}; mkdir -p $O
cd $R
GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/trace.json 2> $O/trace.err; grep "\[open\]" $O/trace.err
