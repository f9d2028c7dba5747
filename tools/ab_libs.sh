# A/B of builds of the library on the headline index: one fresh process per line, alternating (GBWT_HIP_LIB).
# usage: tools/ab_libs.sh TAG [sweep_env configs] [library ...]   (default: libgbwt_hip_prev.so against libgbwt_hip.so)
O=gpurun_out/${1:-ab}; mkdir -p $O
C=$PWD/gbwt_rs_amd/csrc
LIBS="${@:3}"; [ -z "$LIBS" ] && LIBS="libgbwt_hip_prev.so libgbwt_hip.so"
for i in 1 2 3; do
  for lib in $LIBS; do
    echo "## $lib" >> $O/ab.txt
    GBWT_HIP_LIB=$C/$lib timeout 600 python tools/sweep_env.py ${SWEEP_ARGS:-} --reps 6 --configs "${2:-}" >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
