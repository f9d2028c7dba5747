# round 5, first GPU check: the tests touched so far + the bench line + box facts
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
(nproc; free -g; df -h /dev/shm /tmp | tail -2; rocm-smi --showmeminfo vram | tail -4) > $O/box.txt 2>&1; cat $O/box.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "parts_of_rows or fixture_extract or large_query or headline_full_size" > $O/tests1.log 2>&1; tail -5 $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_full_size.py -m gpu -x -q > $O/tests2.log 2>&1; tail -5 $O/tests2.log
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json; tail -5 $O/bench.err
python - <<'P'
import json
j=json.load(open('gpurun_out/r5a/bench.json'))
print(j['value'], j['parity_checked_paths'], j['cpu_baseline'])
print(json.dumps(j['search'],indent=1)[:3000])
print(j['high_degree'].get('cpu_baseline'))
P
