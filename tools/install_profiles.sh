# Copies what tools/measure_round.sh TAG left under gpurun_out/TAG into profiles/rNN_* (the tracked evidence).   usage: install_profiles.sh TAG rNN
O=gpurun_out/$1; P=profiles/$2
cp $O/bench_with_traffic.json ${P}_bench.json; cp $O/bench.json ${P}_bench_first.json; cp $O/hbm_traffic.json ${P}_hbm_traffic.json; cp $O/open_trace.txt ${P}_open_trace.txt
for C in secondary high_degree search config4 config4_small; do
  cp $O/${C}_hbm_traffic.json ${P}_${C}_hbm_traffic.json
  cp "$(find $O/${C}_stats -name '*kernel_stats.csv' | head -1)" ${P}_${C}_kernel_stats.csv
done
cp "$(find $O/stats -name '*kernel_stats.csv' | head -1)" ${P}_kernel_stats.csv
sed -i "s|profiles/$1_|profiles/$2_|g" ${P}_bench.json ${P}_bench_first.json    # (the traffic files the line cites, under the names they are committed with)
python3 - <<P
import json, sys
sys.path.insert(0, '.')
import bench
fp = bench.source_fingerprint()
for n in ['', 'secondary_', 'high_degree_', 'search_', 'config4_', 'config4_small_']:
    t = json.load(open('${P}_%shbm_traffic.json' % n))
    assert t['source_fingerprint'] == fp, (n, t['source_fingerprint'], fp)
print('profiles of build', fp)
P
