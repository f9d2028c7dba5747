# A/B of two builds of the library on the headline index: one fresh process per line, alternating (GBWT_HIP_LIB)
O=gpurun_out/${1:-ab}; mkdir -p $O
P=$PWD/gbwt_rs_amd/csrc/libgbwt_hip_prev.so; N=$PWD/gbwt_rs_amd/csrc/libgbwt_hip.so
for i in 1 2 3; do
  for lib in $P $N; do
    echo "## $(basename $lib)" >> $O/ab.txt
    GBWT_HIP_LIB=$lib timeout 600 python tools/sweep_env.py --reps 6 --configs "${2:-}" >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
