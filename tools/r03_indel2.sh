R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03m}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented or indel or chopped or walk_loop or walker_order or walk_tables" > $O/tests.log 2>&1; tail -3 $O/tests.log
for cu in 1 0; do
echo "CATCH_UP=$cu"
GBWT_HIP_CATCH_UP=$cu timeout 900 python tools/indel_bench.py --extra 1,3 --indel-every 1,8,64,4096 --repeats 4 2>&1 | grep -v amdgpu | cut -c1-260
done > $O/indel.txt 2>&1
cat $O/indel.txt
