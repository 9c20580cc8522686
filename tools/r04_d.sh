#!/bin/bash
# ablation of the free-running 8 / 16-wave attention kernels (mask bits: 1 staging, 2 barrier, 4 LDS fragment reads, 8 exponentials)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04d
mkdir -p "$out"
cd "$root"
P=$root/speech-enhancement-by-s3prl_amd
{
timeout -k 10 100 python3 tools/mhsa_variants.py 0 9 16
for m in 1 2 4 8 3 7 15; do SE_AMD_LIB=$P/libse_amd.abl$m.so timeout -k 10 100 python3 tools/mhsa_variants.py 9 16; done
} > "$out/r04_mhsaN_ablation.txt" 2>&1
cat "$out/r04_mhsaN_ablation.txt"
