#!/bin/bash
# row-complete kernel without its LayerNorm for N = 768, K >= 1536: parity, then the fine-tune step with and without it
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04w
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_gpu_encoder_blocks.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder_train.py tests/test_gpu_spechead_train.py tests/test_gpu_encoder_fp32.py -x -q -m gpu 2>&1 | tail -3
: > "$out/r04_gemm7_plain.txt"
for rep in 1 2; do for v in 0 1; do
SE_AMD_GEMM7_PLAIN=$v timeout -k 10 300 python3 bench.py --workload finetune --no-extras > "$out/ft$v.json" 2> "$out/ft.err" || { tail -20 "$out/ft.err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/ft$v.json').read().strip().splitlines()[-1]); print('finetune SE_AMD_GEMM7_PLAIN=$v', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_gemm7_plain.txt"
done; done
for v in 0 1; do
SE_AMD_GEMM7_PLAIN=$v timeout -k 10 200 python3 tools/x3_pass.py bf16x3 32 5 2>&1 | grep "utt/s" | sed "s/^/SE_AMD_GEMM7_PLAIN=$v /" | tee -a "$out/r04_gemm7_plain.txt"
done
