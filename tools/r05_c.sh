#!/bin/bash
# gemm6q: late-start sweep on the product library
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05c
mkdir -p "$out"
cd "$root"
f="$out/r05_gemm6q_late.txt"
: > "$f"
for rep in 1 2; do for ls in 0 1 2 4 8; do
echo "== SE_AMD_GEMM6P_LATE=$ls" | tee -a "$f"
SE_AMD_GEMM6P_LATE=$ls timeout -k 10 200 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | cut -c1-100 | tee -a "$f"
done; done
