R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ao; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for M in parts paths; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${M}_fetch -- python3 $R/tools/shard_traffic.py $M > $O/${M}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${M}_write -- python3 $R/tools/shard_traffic.py $M > $O/${M}_write.log 2>&1
tail -1 $O/${M}_fetch.log
A=$(tail -1 $O/${M}_fetch.log | awk '{print $2*4}')
cd $R; python3 tools/hbm_traffic.py $O/${M}_fetch $O/${M}_write k_walk_direct "one rank of eight, by $M" $A --key shard_$M --pick last > $O/${M}_traffic.json; cd /tmp
python3 -c "
import json; j=json.load(open('$O/${M}_traffic.json')); print('$M', {k: j[k] for k in ('fetch_bytes_corrected','write_bytes','traffic_bytes_per_launch','algorithmic_bytes_per_launch')})"
done
find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*counter_collection.csv" -size +5M -delete
