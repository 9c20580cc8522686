#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04l
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_gpu_encoder_blocks.py tests/test_gpu_scoring.py tests/test_gpu_encoder_train.py -x -q -m gpu -k "wgrad or scoring or grad" 2>&1 | tail -3
{ for i in 1 2; do for v in 0 1; do echo "== SE_AMD_WGRAD_STAG=$v"; SE_AMD_WGRAD_STAG=$v timeout -k 10 200 python3 tools/bench_kernels.py wgrad 2>&1 | grep -i "wgrad"; done; done
for v in 0 1; do SE_AMD_WGRAD_STAG=$v timeout -k 10 300 python3 bench.py --workload finetune --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('finetune stag=$v', round(d['value']), 'utt/s', round(d['ms_per_step'],3), 'ms')"; done; } | tee "$out/r04l_wgrad_stag.txt"
