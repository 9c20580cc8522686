#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04i
mkdir -p "$out"
cd "$root"
{ for mn in 1536 768 1536 768; do echo "SE_AMD_GEMM6_MIN_N=$mn"; SE_AMD_GEMM6_MIN_N=$mn timeout -k 10 200 python3 tools/bench_kernels.py gemm 2>&1 | grep "N=768"; done; } | tee "$out/r04i_gemm6_n768.txt"
for mn in 1536 768; do SE_AMD_GEMM6_MIN_N=$mn timeout -k 10 300 python3 bench.py --workload finetune --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('finetune min_n=$mn', round(d['value']), 'utt/s', round(d['ms_per_step'],3), 'ms')"; done | tee -a "$out/r04i_gemm6_n768.txt"
