#!/bin/bash
# compile-time ablation of the persistent 256 x 256 GEMM (timing only): what the launch spends on LDS-DMA, barriers, LDS fragment reads, epilogue stores
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04aj
mkdir -p "$out"
cd "$root"
: > "$out/r04_gemm6p_ablation_b.txt"
for rep in 1 2; do for lib in libse_amd.so libse_amd.abl8.so libse_amd.abl16.so; do
echo "== $lib" | tee -a "$out/r04_gemm6p_ablation_b.txt"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 200 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | cut -c1-100 | tee -a "$out/r04_gemm6p_ablation_b.txt"
done; done
