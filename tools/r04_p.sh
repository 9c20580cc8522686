#!/bin/bash
# same-box A/B: packed row-sum adds (product build) vs scalar adds (pk0 build); variants 0 (mhsa.hip), 10 (mhsaN<8,4,1>), 11 (persistent)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04p
mkdir -p "$out"
cd "$root"
for rep in 1 2 3; do
  for lib in libse_amd.so libse_amd.pk0.so; do
    SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 120 python3 tools/mhsa_variants.py 0 10 11 16 2>&1 | grep -v amdgpu.ids | tee -a "$out/r04_mhsa_pkadd_ab.txt"
  done
done
