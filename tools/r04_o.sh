#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04o
mkdir -p "$out"
cd "$root"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/libse_amd.clk.so timeout -k 10 300 python3 tools/clk_probe.py 2>&1 | grep -v amdgpu.ids | tee "$out/r04_clk_probe.txt"
