# One gpurun call that produces everything a round quotes: parity suite (corrupt-file test included), bench line, rocprofv3 kernel stats
# of the bench, and the PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of the headline kernel and of every other config.
# usage: measure_round.sh TAG [quick]
set -x
R=$GRAFT_REPO_ROOT; T=${1:-r06}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
GBWT_HIP_TRACE_OPEN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>&1 >/dev/null | grep "\[open\]\|\[load\]" | tail -24 > $O/open_trace.txt; cat $O/open_trace.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-extras > $O/write.log 2>&1
cd $R
python3 tools/hbm_traffic.py $O/fetch $O/write k_walk_direct "headline C5k (333334 sites x 5000 haplotypes, mosaic, seed 42)" 13333360000 > $O/hbm_traffic.json; head -30 $O/hbm_traffic.json
find $O/stats -name "*kernel_stats.csv" -exec head -12 {} \; | cut -c1-200
if [ "${2:-}" != "quick" ]; then
  for C in secondary high_degree search config4 config4_small; do
    F=""
    case $C in
      search) K=k_search,k_bd_search; P=max; F="--fetch-factor 1";;
      config4|config4_small) K=k_walk_direct,k_format_chunks; P=max;;   # (no request sizes a line: the line cache is filled at open, k_segment_text)
      *) K=k_walk_direct; P=last;;
    esac
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${C}_stats -- python3 $R/tools/configs.py $C > $O/${C}.json 2> $O/${C}.err
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${C}_fetch -- python3 $R/tools/configs.py $C > $O/${C}_fetch.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${C}_write -- python3 $R/tools/configs.py $C > $O/${C}_write.log 2>&1
    cd $R
    A=$(python3 -c "import json,sys; print(json.loads(open('$O/${C}.json').read().strip().splitlines()[-1])['algorithmic_bytes'])")
    python3 tools/hbm_traffic.py $O/${C}_fetch $O/${C}_write $K "$C (tools/configs.py)" $A --key $C --pick $P $F > $O/${C}_hbm_traffic.json; head -12 $O/${C}_hbm_traffic.json
    find $O/${C}_stats -name "*kernel_stats.csv" -exec head -8 {} \; | cut -c1-200
  done
fi
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
# the bench line again, now that the PMC files of THIS build exist (bench.py quotes traffic only from a profile with its own fingerprint)
if [ "${2:-}" != "quick" ]; then
  mkdir -p $R/profiles
  cp $O/hbm_traffic.json $R/profiles/${T}_hbm_traffic.json
  for C in secondary high_degree search config4 config4_small; do cp $O/${C}_hbm_traffic.json $R/profiles/${T}_${C}_hbm_traffic.json; done
  timeout 1200 python bench.py > $O/bench_with_traffic.json 2> $O/bench_with_traffic.err; cut -c1-600 $O/bench_with_traffic.json
fi
