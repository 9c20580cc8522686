#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04ak
mkdir -p "$out"
cd "$root"
: > "$out/r04_gemm6p_skew.txt"
for rep in 1 2; do for v in 0 1 2 4 8; do
echo "== SE_AMD_GEMM6P_SKEW=$v" | tee -a "$out/r04_gemm6p_skew.txt"
SE_AMD_GEMM6P_SKEW=$v timeout -k 10 200 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | cut -c1-100 | tee -a "$out/r04_gemm6p_skew.txt"
done; done
