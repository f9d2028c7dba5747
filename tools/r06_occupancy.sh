# round 6: walkers per CU against config 4's ragged walk and the headline (tools/c4_occupancy_probe.py)
O=gpurun_out/r06d; mkdir -p $O
C=$PWD/gbwt_rs_amd/csrc
for size in small full; do
  for lib in libgbwt_hip.so libgbwt_hip_w5.so libgbwt_hip.so libgbwt_hip_w5.so; do
    GBWT_HIP_LIB=$C/$lib timeout 900 python tools/c4_occupancy_probe.py $size >> $O/occupancy_$size.txt 2>> $O/occupancy.err
  done
  rm -f /dev/shm/gbwt_c4_occ_$size.gbz /dev/shm/gbwt_c4_occ_$size.gbz.generic.npy
  cat $O/occupancy_$size.txt
done
