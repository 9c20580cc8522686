#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r03stft
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_gpu_preprocessor.py -x -q -m gpu > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
timeout -k 10 300 python3 tools/bench_kernels.py stft > "$out/stft.txt" 2>&1 || { tail -20 "$out/stft.txt"; exit 1; }
cat "$out/stft.txt"
