#!/bin/bash
# Counter passes over the walk kernel for several knob settings: tools/pmc_compare.sh OUTDIR "CONFIG;CONFIG;..." "GROUP" "GROUP" ...
# (a CONFIG is what tools/sweep_env.py takes: NAME=VALUE,NAME=VALUE with the GBWT_HIP_ prefix implied; a GROUP is a
# space-separated list of counters that fit one pass).  One rocprofv3 run per (config, group).  SWEEP_ARGS="--extra 1" etc.
# are passed on to sweep_env.py.
set -u
out=$1; configs=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
case "$out" in /*) ;; *) out="$PWD/$out" ;; esac
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra CFGS <<< "$configs"
for cfg in "${CFGS[@]}"; do
    tag=$(echo "${cfg:-defaults}" | tr ',=' '__')
    echo "# config: ${cfg:-(defaults)}"
    i=0
    for group in "$@"; do
        i=$((i + 1))
        d="$out/$tag/pass$i"
        rm -rf "$d"; mkdir -p "$out/$tag"
        # shellcheck disable=SC2086
        timeout --foreground -k 10 120 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$d" -- python3 "$root/tools/sweep_env.py" --reps 2 ${SWEEP_ARGS:-} --configs "$cfg" > "$d.log" 2>&1
        echo "## pass $i: $group"
        python3 "$root/tools/pmc_summary.py" "$d" k_walk
    done
done
