set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O
cd $R
for S in 1 4 8 16; do echo "## GBWT_HIP_UPLOAD_SLICES=$S"; GBWT_HIP_UPLOAD_SLICES=$S timeout 600 python tools/c4_open_trace.py full 2>/dev/null; done > $O/slices.txt; cat $O/slices.txt
