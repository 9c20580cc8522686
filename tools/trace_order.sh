#!/bin/bash
# the ordered kernel list of the LAST pass of `bench.py <args>` (rocprofv3 --kernel-trace): where the glue launches sit
#   tools/trace_order.sh <tag> [bench.py args]
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/tr" -o $tag -- python3 "$root/bench.py" "$@" --steps 2 --warmup 2 --no-roofline --no-cpu-baseline --no-host-fed --no-extras > "$out/tr.log" 2>&1 || { tail -5 "$out/tr.log"; exit 1; }
python3 - "$(find "$out/tr" -name '*kernel_trace.csv' | head -1)" > "$out/${tag}_order.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last pass = from the last stft kernel on
is_stft = lambda n: ('stft_kernel' in n or 'stft_small_kernel' in n) and 'istft' not in n
last = max(i for i, n in enumerate(names) if is_stft(n))
prev = max(i for i, n in enumerate(names[:last]) if is_stft(n))
t0 = int(rows[prev]['Start_Timestamp'])
for r in rows[prev:last]:
    print(f"{(int(r['Start_Timestamp']) - t0)/1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:7.1f} us  {r['Kernel_Name'][:110]}")
PY
rm -rf "$out/tr"
cat "$out/${tag}_order.txt"
