# Waiting for a full ring inside the gather loops: three builds (the uniform loop's wait only; + the gather loops', unbounded; + bounded by the
# caller's patience), alternating fresh processes, per workload: walk min / avg ms
L="libgbwt_hip_uwait.so libgbwt_hip_gwait.so libgbwt_hip.so"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segmented or indel or chopped or chained or walk_tables" 2>&1 | tail -2
i=0
for w in "--extra 3 --indel-every 4096" "--extra 1 --indel-every 64" "--extra 1" "--chop 4 --extra 1"; do
  i=$((i+1)); SWEEP_ARGS="$w" bash tools/ab_libs.sh gp$i "" $L > /dev/null
  echo "#### $w"; grep -v "^$" gpurun_out/gp$i/ab.txt | paste - - | awk '{print $2, $6, $8, "ms"}'
done
