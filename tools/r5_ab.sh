set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ab; mkdir -p $O
cd $R
for i in 1 2 3; do
  for lib in $R/tools/ab_prev/gbwt_rs_amd/csrc/libgbwt_hip.so $R/gbwt_rs_amd/csrc/libgbwt_hip.so; do
    GBWT_HIP_LIB=$lib timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.split('/')[-4], round(j['roofline']['kernel_ms'],3), 'ms', round(j['value']/1e9,1), 'G')" | tee -a $O/ab.txt
  done
done
GBWT_HIP_TRACE_OPEN=1 timeout 900 python tools/c4_open_trace.py full > $O/c4_open.txt 2> $O/c4_open.err; cat $O/c4_open.txt; grep "\[open\]\|\[load\]\|====" $O/c4_open.err | tail -90
