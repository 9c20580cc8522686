#!/bin/bash
# round-4 call H: full GPU test-suite + default bench line on the new default attention kernel
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04h
mkdir -p "$out"
cd "$root"
rm -f gpurun_out/parity_measured.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1 || { tail -30 "$out/pytest_gpu.log"; exit 1; }
tail -3 "$out/pytest_gpu.log"
timeout -k 10 300 python3 bench.py > "$out/r04h_bench.json" 2> "$out/bench.err"
python3 - "$out/r04h_bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('utt/s', round(d['value']), 'ms', round(d['ms_per_step'], 3), 'gemm frac', round(d['roofline']['frac'], 3))
for k, v in d['roofline_other_kernels'].items():
    if isinstance(v, dict): print('  ', k, round(v.get('frac', 0), 3), round(v.get('avg_launch_ms', 0) * 1e3, 1), 'us')
PY
