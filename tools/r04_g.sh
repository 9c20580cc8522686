#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04g
mkdir -p "$out"
cd "$root"
{ for b in 1 2 4 6 8 12 16 24 32 64; do MHSA_B=$b timeout -k 10 100 python3 tools/mhsa_variants.py 0 10; done; MHSA_B=32 MHSA_T=500 timeout -k 10 100 python3 tools/mhsa_variants.py 0 10;  MHSA_B=12 MHSA_T=300 timeout -k 10 100 python3 tools/mhsa_variants.py 0 10; } 2>&1 | grep -v amdgpu.ids | tee "$out/r04g_bsweep.txt"
