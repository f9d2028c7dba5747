O=gpurun_out/${1:-c5d}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python tools/sweep_env.py --sites 3000 --alleles 300 --model iid --reps 5 --configs ";DEEP_TABLES=0;HELPER_NAPS=1;HELPER_NAPS=2;HELPER_NAPS=8;RING_SLOTS=128;DEBUG_DRY_ROWS=64" > $O/probe.txt 2>&1
python tools/sweep_env.py --sites 20000 --alleles 7 --model iid --reps 5 --configs ";DEEP_TABLES=0;HELPER_NAPS=1;HELPER_NAPS=2;HELPER_NAPS=8;RING_SLOTS=128;DEBUG_DRY_ROWS=64" >> $O/probe.txt 2>&1
cat $O/probe.txt
