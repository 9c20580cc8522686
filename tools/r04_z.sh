#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04z
mkdir -p "$out"
cd "$root"
: > "$out/r04_gemm_tailsplit.txt"
for rep in 1 2; do for v in 0 1; do
echo "SE_AMD_GEMM_TAILSPLIT=$v" | tee -a "$out/r04_gemm_tailsplit.txt"
SE_AMD_GEMM_TAILSPLIT=$v timeout -k 10 300 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | tee -a "$out/r04_gemm_tailsplit.txt"
done; done
for v in 0 1 0 1; do
SE_AMD_GEMM_TAILSPLIT=$v timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline > "$out/b$v.json" 2> "$out/err" || { tail -20 "$out/err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/b$v.json').read().strip().splitlines()[-1]); print('enhance SE_AMD_GEMM_TAILSPLIT=$v', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_gemm_tailsplit.txt"
done
