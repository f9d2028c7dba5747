R=$GRAFT_REPO_ROOT; cd $R
run() { env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$*', 'kernel_ms', round(j['roofline']['kernel_ms'],3), 'value', round(j['value']/1e9,1))" 2>/dev/null || echo "$* failed (wrong output by design?)"; }
run A=1
run GBWT_HIP_DEBUG_DRY_ROWS=128
run GBWT_HIP_HELPER_NAPS=2
run GBWT_HIP_HELPER_NAPS=8
run GBWT_HIP_LOOKAHEAD_HOPS=7
run GBWT_HIP_LOOKAHEAD_HOPS=31
timeout 900 python tools/dry_modes.py 2>&1 | grep -v amdgpu
