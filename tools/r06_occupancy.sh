# round 6: walkers per CU against config 4's ragged walk and the headline (tools/c4_occupancy_probe.py)
# libgbwt_hip_w5.so = the same sources with five waves per SIMD asked of the walk kernel (96 VGPRs), built here before the call:
#   cd gbwt_rs_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGBWT_HIP_WALK_WAVES=5 -c -o /tmp/walk_direct_w5.o walk_direct.hip &&
#   hipcc --offload-arch=gfx950 -shared -o libgbwt_hip_w5.so $(ls *.o | grep -v 'walk_direct.o\|comm_testtransport.o') /tmp/walk_direct_w5.o -ldl -lpthread
O=gpurun_out/r06d; mkdir -p $O
C=$PWD/gbwt_rs_amd/csrc
for size in small full; do
  for lib in libgbwt_hip.so libgbwt_hip_w5.so libgbwt_hip.so libgbwt_hip_w5.so; do
    GBWT_HIP_LIB=$C/$lib timeout 900 python tools/c4_occupancy_probe.py $size >> $O/occupancy_$size.txt 2>> $O/occupancy.err
  done
  rm -f /dev/shm/gbwt_c4_occ_$size.gbz /dev/shm/gbwt_c4_occ_$size.gbz.generic.npy
  cat $O/occupancy_$size.txt
done
