#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04k
mkdir -p "$out"
cd "$root"
P=$root/speech-enhancement-by-s3prl_amd
timeout -k 10 300 python3 -m pytest tests/test_gpu_preprocessor.py -x -q -m gpu 2>&1 | tail -2
{ for i in 1 2; do for v in stftold new; do if [ $v = new ]; then unset SE_AMD_LIB; else export SE_AMD_LIB=$P/libse_amd.$v.so; fi; echo "== $v"; timeout -k 10 200 python3 tools/bench_kernels.py stft 2>&1 | grep "stft " ; done; done; } | tee "$out/r04k_stft_ab.txt"
