R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_am; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rows_sized or parts_of_rows or segmented or ragged or headline or output" > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-600
timeout 600 python tools/slab_probe.py 2>&1 | grep -v amdgpu | grep "N=1 \|N=8"
GBWT_HIP_DEFER_TOTAL=0 timeout 600 python tools/slab_probe.py 2>&1 | grep -v amdgpu | grep "N=1 \|N=8"
