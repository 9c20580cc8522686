#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py <args>`; the summary lands in gpurun_out/<tag>/<tag>_kernel_stats.csv
#   tools/prof_stats.sh <tag> [bench.py args...]
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o $tag -- python3 "$root/bench.py" "$@" --no-cpu-baseline --no-host-fed --no-extras > "$out/prof.log" 2>&1 || { tail -20 "$out/prof.log"; exit 1; }
cp $(find "$out/prof" -name "*kernel_stats.csv" | head -1) "$out/${tag}_kernel_stats.csv"
rm -rf "$out/prof"
python3 - "$out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
