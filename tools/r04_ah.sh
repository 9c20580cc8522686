#!/bin/bash
# wgrad STAG=3 (asm fragment reads, software-pipelined across sub-steps) vs the staggered default
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04ah
mkdir -p "$out"
cd "$root"
SE_AMD_WGRAD_STAG=3 timeout -k 10 600 python3 -m pytest tests/test_gpu_spechead_train.py tests/test_gpu_scoring.py tests/test_gpu_encoder_train.py -x -q -m gpu 2>&1 | tail -3
: > "$out/r04_wgrad_stag3.txt"
for rep in 1 2; do for v in 1 3; do
  echo "== SE_AMD_WGRAD_STAG=$v" | tee -a "$out/r04_wgrad_stag3.txt"
  SE_AMD_WGRAD_STAG=$v timeout -k 10 300 python3 tools/bench_kernels.py wgrad 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a "$out/r04_wgrad_stag3.txt"
done; done
for v in 1 3 1 3; do
SE_AMD_WGRAD_STAG=$v timeout -k 10 300 python3 bench.py --workload finetune --no-extras > "$out/ft$v.json" 2> "$out/ft.err" || { tail -20 "$out/ft.err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/ft$v.json').read().strip().splitlines()[-1]); print('finetune stag=$v', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_wgrad_stag3.txt"
done
