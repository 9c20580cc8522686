#!/bin/bash
# LSTM kernels with inline-asm LDS-DMA + lgkmcnt-only step barrier: parity tests, then the LSTM-head training step
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04t
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_gpu_lstm.py tests/test_gpu_scoring.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python3 bench.py --workload lstm --no-extras > "$out/lstm.json" 2> "$out/lstm.err" || { tail -20 "$out/lstm.err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/lstm.json').read().strip().splitlines()[-1]); print('lstm', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee "$out/r04_lstm_asm_dma.txt"
