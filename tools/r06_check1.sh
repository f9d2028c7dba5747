# round 6, first GPU check: the line cache filled at open, the lean-handle refusal, the all-path oracle parity
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gfa.py -x -q -k "not c4_full_size" > $O/gfa_tests.log 2>&1; tail -5 $O/gfa_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lean or open_flags or every_knob or knobs" > $O/parity_tests.log 2>&1; tail -5 $O/parity_tests.log
GBWT_HIP_TRACE_OPEN=1 timeout 900 python tools/c4_bench.py --size full --passes 3 > $O/c4_full.json 2> $O/c4_full.err; cat $O/c4_full.json | cut -c1-3000; grep "\[open\]" $O/c4_full.err | tail -30
timeout 600 python tools/c4_bench.py --size small --passes 3 > $O/c4_small.json 2> $O/c4_small.err; cut -c1-1500 $O/c4_small.json
