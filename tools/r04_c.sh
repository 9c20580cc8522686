#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04c
mkdir -p "$out"
cd "$root"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/libse_amd.stamps.so timeout -k 10 200 python3 tools/mhsa_stamps.py > "$out/r04_mhsa_stamps.txt" 2>&1
tail -32 "$out/r04_mhsa_stamps.txt"
