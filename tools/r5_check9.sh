set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
timeout 900 python tools/c4_knob_sweep.py full > $O/c4_knob_sweep_full.txt 2> $O/sweep.err; cat $O/c4_knob_sweep_full.txt; tail -3 $O/sweep.err
timeout 900 python tools/c4_knob_sweep.py small > $O/c4_knob_sweep_small.txt 2> $O/sweep2.err; cat $O/c4_knob_sweep_small.txt; tail -3 $O/sweep2.err
