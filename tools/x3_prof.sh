#!/bin/bash
# kernel stats of the bf16x3 parity-mode pass -> gpurun_out/x3/x3_kernel_stats.csv
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/x3
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o x3 -- python3 "$root/tools/x3_pass.py" ${1:-bf16x3} ${2:-32} 5 > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
cp $(find "$out/prof" -name "*kernel_stats.csv" | head -1) "$out/x3_kernel_stats.csv"
rm -rf "$out/prof"
grep "utt/s" "$out/prof.log"
python3 - "$out/x3_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
