#!/bin/bash
# round 5, first contact of gemm6q with the hardware: its parity tests, then an interleaved A/B against gemm6p on the bench shapes
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05a
mkdir -p "$out"
cd "$root"
timeout -k 10 400 python3 -m pytest tests/test_gpu_gemm6q.py -x -q 2>&1 | tail -15 | tee "$out/tests.txt"
f="$out/r05_gemm6q_ab.txt"
: > "$f"
for rep in 1 2 3; do for q in 0 1; do
echo "== SE_AMD_GEMM6Q=$q" | tee -a "$f"
SE_AMD_GEMM6Q=$q timeout -k 10 200 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | cut -c1-100 | tee -a "$f"
done; done
