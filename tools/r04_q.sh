#!/bin/bash
# full GPU suite + default bench + x3 parity-mode timing
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04q
mkdir -p "$out"
cd "$root"
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1 || { tail -40 "$out/pytest_gpu.log"; exit 1; }
tail -3 "$out/pytest_gpu.log"
timeout -k 10 300 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
python3 - "$out/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms', 'roofline', d['roofline']['frac'], {k: v for k, v in d.items() if 'parity' in k or 'host_fed' in k})
PY
