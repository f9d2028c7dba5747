#!/bin/bash
# The other BASELINE configurations on one GPU (C2, i.i.d. stress, C5 high degree, multi-allelic, ragged rows, C3 search, GFA lines,
# insertion chains):
# walk / search kernel times, and for C5 the rocprofv3 kernel stats + HBM traffic (FETCH_SIZE, WRITE_SIZE in separate --pmc
# passes, FETCH doubled per the gfx950 correction).   usage: tools/configs_round.sh OUTDIR
set -u
out=$1
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"; case "$out" in /*) ;; *) out="$PWD/$out" ;; esac
cd "$root"
run() { echo "## $*"; timeout 600 python3 "$@" 2>&1 | grep -v "amdgpu.ids"; }
{
echo "# Round 3, 1 x MI355X: the other configurations (walk-kernel time from HIP events; 'M steps/s' of sweep.py are G LF-steps/s)"
run tools/sweep.py --sites 3333 --haplotypes 1000 --model mosaic --configs 0:64:16 --reps 3
run tools/sweep.py --sites 3333 --haplotypes 1000 --model iid --configs 0:64:16 --reps 3
run tools/sweep.py --sites 100000 --model mosaic --configs 0:64:16 --reps 3
run tools/sweep.py --sites 100000 --model iid --configs 0:64:16 --reps 3
run tools/sweep.py --sites 20000 --alleles 7 --model iid --configs 0:64:16 --reps 3
run tools/sweep.py --sites 3000 --alleles 300 --model iid --configs 0:64:16 --reps 3
run tools/sweep.py --sites 3000 --alleles 400 --model iid --configs 0:64:16 --reps 3
run tools/ragged_bench.py
run tools/search_bench.py --sites 1100000 --haplotypes 5008
run tools/gfa_bench.py --sites 20000
run tools/indel_bench.py --extra 0,1,3 --indel-every 1,8,64,4096 --repeats 3
run tools/indel_bench.py --extra 0,1 --repeats 3 --chop 4
} > "$out/other_configs.txt"
# C5 under rocprofv3: kernel stats, then the two traffic passes
cd /tmp && export TMPDIR=/tmp
C5="$root/tools/sweep.py --sites 3000 --alleles 300 --model iid --configs 0:64:16 --reps 3"
timeout --foreground 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c5_stats" -- python3 $C5 > "$out/c5_stats.log" 2>&1
timeout --foreground 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/c5_fetch" -- python3 $C5 > "$out/c5_fetch.log" 2>&1
timeout --foreground 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/c5_write" -- python3 $C5 > "$out/c5_write.log" 2>&1
cd "$root"
python3 - "$out" <<'PY' >> "$out/other_configs.txt"
import csv, glob, sys
out = sys.argv[1]
def rows(d, name):
    for p in glob.glob(f"{out}/{d}/**/*{name}.csv", recursive=True):
        yield from csv.DictReader(open(p))
print("\n## C5 (3 000 sites x 300 alleles, Zipf 1.2, 5 000 haplotypes) under rocprofv3: k_walk_direct")
dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows("c5_stats", "kernel_trace") if "k_walk_direct" in r["Kernel_Name"]]
def counter(d, c):
    per = {}
    for r in rows(d, "counter_collection"):
        if "k_walk_direct" in r["Kernel_Name"] and r["Counter_Name"] == c:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return [per[k] for k in sorted(per, key=int)]
f, w = counter("c5_fetch", "FETCH_SIZE"), counter("c5_write", "WRITE_SIZE")
if dur and f and w:
    t = min(dur[1:] or dur) * 1e-9
    fb, wb = 2 * f[-1] * 1024, w[-1] * 1024
    print(f"launches {len(dur)}, duration min {min(dur) / 1e6:.3f} ms avg {sum(dur) / len(dur) / 1e6:.3f} ms")
    print(f"HBM traffic per launch: fetch {fb / 1e9:.3f} GB (2 x FETCH_SIZE), write {wb / 1e9:.3f} GB -> {(fb + wb) / t / 1e9:.0f} GB/s = {(fb + wb) / t / 8e12:.3f} of 8 TB/s")
else:
    print("profile incomplete:", len(dur), len(f), len(w))
PY
cat "$out/other_configs.txt"
