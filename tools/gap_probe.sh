# Checkpoint gap (LF steps per checkpoint record) against the default of half the sample interval: walk-kernel times per workload
O=gpurun_out/${1:-gap}; mkdir -p $O
CFG=";CHECKPOINT_GAP=512;CHECKPOINT_GAP=256;CHECKPOINT_GAP=128;CHECKPOINT_GAP=256,CATCH_UP=0;"
while read -r w; do
  echo "## $w" >> $O/gap.txt
  timeout 900 python tools/sweep_env.py $w --reps 4 --configs "$CFG" >> $O/gap.txt 2>&1
done <<'L'

--extra 1 --indel-every 64
--extra 1 --indel-every 8
--extra 3
--chop 4
--chop 4 --extra 1
--sites 100000 --model iid
L
cat $O/gap.txt
