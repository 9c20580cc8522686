#!/bin/bash
# re-validation after the stale-library episode: LSTM (asm DMA), plain gemm7, x3, on the library actually built from this tree
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04x
mkdir -p "$out"
cd "$root"
python3 - <<'PY'
import ctypes, sys
sys.path.insert(0, '.')
from speech_enhancement_by_s3prl_amd import _lib
lib = _lib.load()
assert hasattr(lib, 'se_gemm7_plain_launch'), 'stale library'
print('library has se_gemm7_plain_launch: fresh')
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_lstm.py tests/test_gpu_scoring.py tests/test_gpu_encoder_blocks.py -x -q -m gpu 2>&1 | tail -3
: > "$out/r04_revalidate.txt"
timeout -k 10 300 python3 bench.py --workload lstm --no-extras > "$out/lstm.json" 2> "$out/err" || { tail -20 "$out/err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/lstm.json').read().strip().splitlines()[-1]); print('lstm (inline-asm DMA)', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_revalidate.txt"
for rep in 1 2; do for v in 0 1; do
SE_AMD_GEMM7_PLAIN=$v timeout -k 10 300 python3 bench.py --workload finetune --no-extras > "$out/ft$v.json" 2> "$out/err" || { tail -20 "$out/err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/ft$v.json').read().strip().splitlines()[-1]); print('finetune SE_AMD_GEMM7_PLAIN=$v', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_revalidate.txt"
done; done
for v in 0 1; do
SE_AMD_GEMM7_PLAIN=$v timeout -k 10 200 python3 tools/x3_pass.py bf16x3 32 5 2>&1 | grep "utt/s" | sed "s/^/SE_AMD_GEMM7_PLAIN=$v /" | tee -a "$out/r04_revalidate.txt"
done
