set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03a}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_corrupt_files_never_take_the_device_down > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 600 python tools/open_bench.py > $O/open_headline.txt 2>&1; cat $O/open_headline.txt
timeout 600 python tools/open_bench.py --sites 666667 --haplotypes 90 > $O/open_c4.txt 2>&1; cat $O/open_c4.txt
