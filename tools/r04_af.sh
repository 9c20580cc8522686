#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04af
mkdir -p "$out"
cd "$root"
: > "$out/r04_gemm7_plain_mink.txt"
for rep in 1 2; do for v in 1536 768; do
SE_AMD_GEMM7_PLAIN_MINK=$v timeout -k 10 300 python3 bench.py --workload finetune --no-extras > "$out/ft.json" 2> "$out/err" || { tail -20 "$out/err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/ft.json').read().strip().splitlines()[-1]); print('finetune SE_AMD_GEMM7_PLAIN_MINK=$v', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_gemm7_plain_mink.txt"
done; done
