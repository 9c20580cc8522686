#!/bin/bash
# round-4 call B: parity tests of the attention variants, then the side-by-side timing
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04b
mkdir -p "$out"
cd "$root"
timeout -k 10 400 python3 -m pytest tests/test_gpu_encoder_blocks.py -x -q -m gpu -k "mhsa_prescaled" > "$out/pytest_mhsa.log" 2>&1 || { tail -30 "$out/pytest_mhsa.log"; exit 1; }
tail -3 "$out/pytest_mhsa.log"
timeout -k 10 200 python3 tools/bench_kernels.py mhsa > "$out/r04b_mhsa.txt" 2>&1
cat "$out/r04b_mhsa.txt"
