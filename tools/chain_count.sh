GBWT_HIP_TRACE_OPEN=1 timeout 900 python tools/indel_bench.py --extra 0,1 --indel-every 1,8,64 --repeats 1 2>&1 | grep "chained\|haplotypes"
GBWT_HIP_TRACE_OPEN=1 timeout 900 python tools/indel_bench.py --extra 0 --chop 4 --repeats 1 2>&1 | grep "chained\|haplotypes"
