R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03l}; mkdir -p $O
cd $R
for cap in "" 100000; do
for serial in "" 1; do
echo "CHECKPOINT_CAP=$cap SERIAL_SAMPLES=$serial"
env ${cap:+GBWT_HIP_CHECKPOINT_CAP=$cap} ${serial:+GBWT_HIP_SERIAL_SAMPLES=$serial} timeout 900 python tools/indel_bench.py --extra 1 --indel-every 1,64,4096 --repeats 4 2>&1 | grep -v amdgpu | cut -c1-260
done; done > $O/indel.txt 2>&1
cat $O/indel.txt
