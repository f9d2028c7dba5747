R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03t}; mkdir -p $O
cd $R
{
for combo in "1:0.5" "1.5:0.75" "1:1" "1.5:1.5" "2:0.5"; do
capf=${combo%%:*}; gapf=${combo##*:}
for cfg in "512 --sites 100000 --model iid" "512 --sites 100000 --model mosaic" "64 --sites 3000 --alleles 300 --model iid" "128 --sites 20000 --alleles 7 --model iid"; do
iv=${cfg%% *}; rest=${cfg#* }
cap=$(python3 -c "print(int($iv*$capf))"); gap=$(python3 -c "print($iv*$gapf)")
echo "## cap=${capf}I gap=${gapf}I ($cap, $gap) $rest"
GBWT_HIP_CHECKPOINT_CAP=$cap GBWT_HIP_CHECKPOINT_GAP=$gap timeout 600 python tools/sweep.py $rest --configs 0:64:16 --reps 5 2>&1 | grep -v amdgpu
done
cap=$(python3 -c "print(int(2048*$capf))"); gap=$(python3 -c "print(2048*$gapf)")
echo "## cap=${capf}I gap=${gapf}I headline + indels"
GBWT_HIP_CHECKPOINT_CAP=$cap GBWT_HIP_CHECKPOINT_GAP=$gap timeout 900 python tools/indel_bench.py --extra 0,1 --indel-every 64,4096 --repeats 4 2>&1 | grep -v amdgpu | cut -c1-60,160-260
done
} > $O/sweep2.txt 2>&1; cat $O/sweep2.txt
