#!/bin/bash
# Counter passes over the walk kernel: one rocprofv3 --pmc run per group (hardware counters of one block are few),
# each over `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $BENCH_ARGS`; prints per-kernel averages.
# usage: tools/pmc_passes.sh OUTDIR "GROUP1" "GROUP2" ...      (a group = space-separated counter names)
set -u
out=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "$@"; do
    i=$((i + 1))
    d="$out/pass$i"
    rm -rf "$d"
    # shellcheck disable=SC2086
    timeout --foreground 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$d" -- python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$d.log" 2>&1
    echo "## pass $i: $group"
    python3 "$root/tools/pmc_summary.py" "$d" k_walk
done
