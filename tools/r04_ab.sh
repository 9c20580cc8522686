#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04ab
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder_fp32.py -x -q -m gpu 2>&1 | tail -3
: > "$out/r04_x3_rowln_split.txt"
for rep in 1 2; do for v in 3 7; do
  SE_AMD_X3_ROWLN=$v timeout -k 10 200 python3 tools/x3_pass.py bf16x3 32 5 2>&1 | grep "utt/s" | sed "s/^/SE_AMD_X3_ROWLN=$v /" | tee -a "$out/r04_x3_rowln_split.txt"
done; done
SE_AMD_X3_ROWLN=7 timeout -k 10 200 python3 tools/x3_pass.py bf16x3 8 5 2>&1 | grep "utt/s" | tee -a "$out/r04_x3_rowln_split.txt"
