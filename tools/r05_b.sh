#!/bin/bash
# compile-time ablation of gemm6q (timing only): tagged libraries given as arguments
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05b
mkdir -p "$out"
cd "$root"
tags=${@:-"q1 q2"}
f="$out/r05_gemm6q_ablation_$(echo $tags | tr ' ' '_').txt"
: > "$f"
for rep in 1 2 3; do for lib in libse_amd.so $(for t in $tags; do echo libse_amd.$t.so; done); do
echo "== $lib" | tee -a "$f"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 200 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | cut -c1-100 | tee -a "$f"
done; done
