#!/bin/bash
# gemm6q placement plans (tagged libraries given as arguments) against gemm6p, same box
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05d
mkdir -p "$out"
cd "$root"
f="$out/r05_gemm6q_plans_$(echo $@ | tr ' ' '_').txt"
: > "$f"
for lib in libse_amd.so $(for t in $@; do echo libse_amd.$t.so; done); do
echo "== $lib" | tee -a "$f"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 200 python3 tools/bench_gemm6q.py 3 2>&1 | grep -v amdgpu.ids | tee -a "$f"
done
