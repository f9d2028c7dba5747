# Chained steps (GBWT_HIP_CHAINS): parity of the chopped / indel cases first, then the walk-kernel times with and without, alternating
O=gpurun_out/${1:-chain}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chopped or indel or segmented or walk_tables" > $O/tests.log 2>&1; tail -5 $O/tests.log
AB=";CHAINS=0;;CHAINS=0"
while read -r w; do
  echo "## $w" >> $O/bench.txt
  timeout 900 python tools/sweep_env.py $w --reps 4 --configs "$AB" >> $O/bench.txt 2>&1
done <<'L'

--extra 1
--extra 3
--extra 1 --indel-every 8
--extra 1 --indel-every 64
--extra 3 --indel-every 4096
--chop 4
--chop 4 --extra 1
L
cat $O/bench.txt
bash tools/ab_libs.sh $1
