#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04ag
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_gpu_objectives.py tests/test_gpu_mix_score.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o head -- python3 "$root/bench.py" --workload head --head-feat linear201 --no-extras --no-cpu-baseline --steps 10 --warmup 2 > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
cp $(find "$out/prof" -name "*kernel_stats.csv" | head -1) "$out/head_kernel_stats.csv"
rm -rf "$out/prof"
grep -o '"value": [0-9.]*' "$out/prof.log" | head -1
python3 - "$out/head_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
