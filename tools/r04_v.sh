#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04v
mkdir -p "$out"
cd "$root"
: > "$out/r04_x3_rowln.txt"
for v in 0 1 2 3; do
  echo "SE_AMD_X3_ROWLN=$v" | tee -a "$out/r04_x3_rowln.txt"
  SE_AMD_X3_ROWLN=$v timeout -k 10 200 python3 tools/x3_pass.py bf16x3 32 5 2>&1 | grep "utt/s" | tee -a "$out/r04_x3_rowln.txt"
done
SE_AMD_X3_ROWLN=3 timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder_fp32.py -x -q -m gpu 2>&1 | tail -4
