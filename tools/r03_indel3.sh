R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented or indel or chopped or walk_loop or walker_order or walk_tables or chain_indexes or high_degree or generic_records" > $O/tests.log 2>&1; tail -3 $O/tests.log
echo "CATCH_UP default"
timeout 900 python tools/indel_bench.py --extra 1,3 --indel-every 1,8,64,4096 --repeats 4 2>&1 | grep -v amdgpu | cut -c1-260 > $O/indel.txt; cat $O/indel.txt
bash tools/r03_c5.sh $1
