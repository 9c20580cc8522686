#!/bin/bash
# FFN1 forward of the training path: pre-activation + GELU from one launch (SE_AMD_GEMM6_DUAL=1) against two launches; parity first
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04y
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder_train.py tests/test_gpu_dropin_sequence.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
: > "$out/r04_gemm6_dual.txt"
for rep in 1 2; do for v in 0 1; do
SE_AMD_GEMM6_DUAL=$v timeout -k 10 300 python3 bench.py --workload finetune --no-extras > "$out/ft$v.json" 2> "$out/err" || { tail -20 "$out/err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/ft$v.json').read().strip().splitlines()[-1]); print('finetune SE_AMD_GEMM6_DUAL=$v', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_gemm6_dual.txt"
done; done
