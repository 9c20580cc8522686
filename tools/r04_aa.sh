#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04aa
mkdir -p "$out"
cd "$root"
timeout -k 10 800 python3 -m pytest tests/test_gpu_encoder_train.py tests/test_gpu_dist_finetune.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python3 tools/bench_kernels.py mhsa_train 2>&1 | grep -v amdgpu.ids | tee "$out/r04_mhsa_train_dpp.txt"
timeout -k 10 300 python3 bench.py --workload finetune --no-extras > "$out/ft.json" 2> "$out/err" || { tail -20 "$out/err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/ft.json').read().strip().splitlines()[-1]); print('finetune', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_mhsa_train_dpp.txt"
