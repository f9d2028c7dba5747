set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
GBWT_HIP_TRACE_GFA=1 timeout 900 python tools/gfa_writers_sweep.py 3 8 16 32 > $O/writers.txt 2> $O/writers.err; cat $O/writers.txt; grep "\[gfa\]" $O/writers.err | tail -12
