#!/bin/bash
# round-3 baseline on today's pool: GPU tests, default bench line, kernel stats
set -e
tag=$1
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1 || { tail -30 "$out/pytest_gpu.log"; exit 1; }
tail -3 "$out/pytest_gpu.log"
timeout -k 10 300 python3 bench.py > "$out/${tag}_bench.json" 2> "$out/bench.err"
tail -c 1500 "$out/${tag}_bench.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o $tag -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed > "$out/prof.log" 2>&1
cp $(find "$out/prof" -name "*kernel_stats.csv" | head -1) "$out/${tag}_bench_kernel_stats.csv"
rm -rf "$out/prof"
