timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented or indel or chopped or chained or walk_tables" 2>&1 | tail -2
SWEEP_ARGS="--extra 1" bash tools/ab_libs.sh gw1 "" > /dev/null; SWEEP_ARGS="--extra 1 --indel-every 8" bash tools/ab_libs.sh gw2 "" >/dev/null; SWEEP_ARGS="--chop 4 --extra 1" bash tools/ab_libs.sh gw3 "" > /dev/null; SWEEP_ARGS="--extra 1" bash tools/ab_libs.sh gw4 "CHAINS=0" > /dev/null
for t in gw1 gw2 gw3 gw4; do echo "#### $t"; grep -v "^$" gpurun_out/$t/ab.txt | paste - - | awk '{print $2, $6, $8, $14}' ; done
