R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03s}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented or indel or chopped or walk_loop or walker_order or walk_tables or chain_indexes or output_paths or headline" > $O/tests.log 2>&1; tail -3 $O/tests.log
{
for cfg in "--sites 100000 --model mosaic" "--sites 100000 --model iid" "--sites 20000 --alleles 7 --model iid" "--sites 3000 --alleles 300 --model iid" "--sites 3333 --haplotypes 1000 --model mosaic"; do
for ser in "" 1; do
echo "## $cfg serial=$ser"; env ${ser:+GBWT_HIP_SERIAL_SAMPLES=1} timeout 600 python tools/sweep.py $cfg --configs 0:64:16 --reps 5 2>&1 | grep -v amdgpu
done; done
timeout 600 python tools/open_bench.py --reps 1 --passes 8 --modes "checkpoint,serial,checkpoint" 2>&1 | grep -v amdgpu | cut -c1-330
timeout 600 python tools/open_bench.py --sites 666667 --haplotypes 90 --reps 1 --modes checkpoint,serial 2>&1 | grep -v amdgpu | cut -c1-330
timeout 900 python tools/indel_bench.py --extra 0,1 --indel-every 1,64,4096 --repeats 3 2>&1 | grep -v amdgpu | cut -c1-260
timeout 900 python tools/indel_bench.py --extra 0 --repeats 3 --chop 4 2>&1 | grep -v amdgpu | cut -c1-260
} > $O/mid.txt 2>&1; cat $O/mid.txt
