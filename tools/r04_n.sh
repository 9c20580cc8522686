#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04n
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder_fp32.py -x -q -m gpu 2>&1 | tail -4
timeout -k 10 200 python3 tools/x3_pass.py bf16x3 32 5 2>&1 | grep "utt/s" | tee "$out/r04n_x3.txt"
timeout -k 10 200 python3 tools/x3_pass.py bf16x3 8 5 2>&1 | grep "utt/s" | tee -a "$out/r04n_x3.txt"
