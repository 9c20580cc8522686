#!/bin/bash
# compile-time ablation of the row-complete GEMM + residual + LayerNorm (gemm7_res_ln_kernel), 24-bit stream form as in the bench: K = 768 and K = 3072
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04al
mkdir -p "$out"
cd "$root"
: > "$out/r04_gemm7_ablation_b.txt"
for rep in 1 2; do for lib in libse_amd.so libse_amd.g7abl4.so libse_amd.g7abl31.so; do
echo "== $lib" | tee -a "$out/r04_gemm7_ablation_b.txt"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 200 python3 tools/bench_kernels.py res24 2>&1 | grep -v amdgpu.ids | grep "variant 7" | cut -c1-100 | tee -a "$out/r04_gemm7_ablation_b.txt"
done; done
