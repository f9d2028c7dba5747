R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03c}; mkdir -p $O
cd $R
timeout 900 python tools/open_bench.py --reps 1 --passes 8 --modes "serial,checkpoint,checkpoint:CHECKPOINT_CAP=2048,checkpoint:CHECKPOINT_CAP=3072,checkpoint:SAMPLE_INTERVAL=3072:CHECKPOINT_CAP=3072,checkpoint:SAMPLE_INTERVAL=4096:CHECKPOINT_CAP=4096,checkpoint:SAMPLE_INTERVAL=1024,serial:SAMPLE_INTERVAL=1600,serial,checkpoint" > $O/exp.txt 2>&1; cat $O/exp.txt
