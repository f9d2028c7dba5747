R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q -k "table or degree or c5 or segmented or chain_indexes or variants or random or rle" > $O/tests.log 2>&1; tail -3 $O/tests.log
for C in 1 0 1 0; do
GBWT_HIP_COMPACT_TABLES=$C timeout 600 python tools/configs.py high_degree 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('compact $C high_degree kernel_ms', j['kernel_ms'], 'G/s', j['value_kernel']/1e9, j['value']/1e9)"
done
