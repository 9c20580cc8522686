#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04ai
mkdir -p "$out"
cd "$root"
: > "$out/r04_gemm6_samesrc_ablation.txt"
for rep in 1 2; do for lib in libse_amd.so libse_amd.abl.so; do
echo "== $lib" | tee -a "$out/r04_gemm6_samesrc_ablation.txt"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 300 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | tee -a "$out/r04_gemm6_samesrc_ablation.txt"
done; done
