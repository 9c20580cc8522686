#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04ac
mkdir -p "$out"
cd "$root"
: > "$out/r04_x3_rowln_batch.txt"
for b in 8 16 24 32; do for v in 0 3 7; do
  SE_AMD_X3_ROWLN=$v timeout -k 10 200 python3 tools/x3_pass.py bf16x3 $b 5 2>&1 | grep "utt/s" | sed "s/^/SE_AMD_X3_ROWLN=$v /" | tee -a "$out/r04_x3_rowln_batch.txt"
done; done
