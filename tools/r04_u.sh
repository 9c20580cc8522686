#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04u
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o lstm -- python3 "$root/bench.py" --workload lstm --no-extras --steps 10 --warmup 2 > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
cp $(find "$out/prof" -name "*kernel_stats.csv" | head -1) "$out/lstm_kernel_stats.csv"
rm -rf "$out/prof"
python3 - "$out/lstm_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
