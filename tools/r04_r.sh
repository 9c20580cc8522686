#!/bin/bash
# wgrad with inline-asm LDS-DMA (no compiler vmcnt(0)): parity tests, kernel timing, fine-tune step
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04r
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_gpu_spechead_train.py tests/test_gpu_scoring.py tests/test_gpu_encoder_train.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python3 tools/bench_kernels.py wgrad 2>&1 | grep -v amdgpu.ids | tee "$out/r04_wgrad_asm_dma.txt"
timeout -k 10 300 python3 bench.py --workload finetune --no-extras > "$out/ft.json" 2> "$out/ft.err" || { tail -20 "$out/ft.err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/ft.json').read().strip().splitlines()[-1]); print('finetune', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_wgrad_asm_dma.txt"
