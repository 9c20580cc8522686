#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04m
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder_blocks.py tests/test_gpu_encoder_pipeline.py tests/test_gpu_fullsize_properties.py tests/test_gpu_graph.py tests/test_gpu_dropin_sequence.py -x -q -m gpu 2>&1 | tail -3
{ for i in 1 2; do for v in 0 1; do echo "== SE_AMD_GEMM7_RLATE=$v"; SE_AMD_GEMM7_RLATE=$v timeout -k 10 200 python3 tools/bench_kernels.py res24 2>&1 | grep -i "variant 7"; done; done
for v in 0 1 0 1; do SE_AMD_GEMM7_RLATE=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('enhance rlate=$v', round(d['value']), 'utt/s', round(d['ms_per_step'],4), 'ms  gemm frac', round(d['roofline']['frac'],4))"; done; } | tee "$out/r04m_rlate.txt"
