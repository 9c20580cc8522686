#!/bin/bash
# Collect the per-round evidence set for profiles/ on the GPU box (everything lands in gpurun_out/<tag>/, which gpurun merges back):
#   <tag>_bench.json               the JSON line of the default `python bench.py` (un-profiled, with cpu_baseline)
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed`
#   <tag>_pmc_fetch_write.json     separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --steps 3 --warmup 1 --no-roofline ...`
# The profiled program runs directly after `--` (no env / bash -c hop: the profiler's preload has initialised the GPU by then).
#   tools/collect_profiles.sh r02                                     the default workload
#   tools/collect_profiles.sh r03f _head_mel120 --workload head       another workload: files <tag><suffix>_bench.json, ... (round 3)
set -e
tag=$1
sfx=$2
shift; [ $# -gt 0 ] && shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
python3 bench.py "$@" > "$out/${tag}${sfx}_bench.json" 2> "$out/bench${sfx}.err"
tail -c 300 "$out/${tag}${sfx}_bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o $tag -- python3 "$root/bench.py" "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed --no-extras > "$out/prof${sfx}.log" 2>&1
cp $(find "$out/prof" -name "*kernel_stats.csv" | head -1) "$out/${tag}${sfx}_bench_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o $c -- python3 "$root/bench.py" "$@" --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-host-fed --no-extras > "$out/pmc_$c${sfx}.log" 2>&1
done
python3 "$root/tools/pmc_summary.py" "$out/${tag}${sfx}_pmc_fetch_write.json" $(find "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" -name '*counter_collection.csv')
rm -rf "$out/prof" "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE"
