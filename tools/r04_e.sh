#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04e
mkdir -p "$out"
cd "$root"
timeout -k 10 400 python3 -m pytest tests/test_gpu_encoder_blocks.py -x -q -m gpu -k "mhsa_prescaled" > "$out/pytest_mhsa.log" 2>&1 || { tail -30 "$out/pytest_mhsa.log"; exit 1; }
tail -3 "$out/pytest_mhsa.log"
{ timeout -k 10 100 python3 tools/mhsa_variants.py 0 9 10 16; SE_AMD_MHSA_STAG=0 timeout -k 10 100 python3 tools/mhsa_variants.py 0 10 16; } 2>&1 | grep -v amdgpu.ids | tee "$out/r04e_variants.txt"
