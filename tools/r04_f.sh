#!/bin/bash
# round-4 call F: head kernel on the three-term bf16 split -- parity tests of everything that calls it, then the configs[3] bench line, old vs new
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04f
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_gpu_heads_decode.py tests/test_gpu_objectives.py tests/test_gpu_train_step.py tests/test_gpu_lstm.py tests/test_gpu_scoring.py tests/test_gpu_fullsize_properties.py tests/test_gpu_graph.py tests/test_gpu_dropin_sequence.py -x -q -m gpu > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
for feat in mel120 linear201; do
  for old in 0 1; do
    if [ $old = 1 ]; then export SE_AMD_HEAD_F32MFMA=1; else unset SE_AMD_HEAD_F32MFMA; fi
    timeout -k 10 300 python3 bench.py --workload head --head-feat $feat --no-cpu-baseline --no-host-fed > "$out/head_${feat}_old$old.json" 2> "$out/head_${feat}_old$old.err"
    python3 - "$out/head_${feat}_old$old.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
hk = d['roofline_other_kernels'].get('head_kernel', {})
print(sys.argv[1].split('/')[-1], 'utt/s', round(d['value']), 'ms', round(d['ms_per_step'], 3), 'head_kernel us', round(hk.get('avg_launch_ms', 0) * 1e3, 1), 'frac', round(hk.get('frac', 0), 3), 'batch12', round(d.get('batch_12', {}).get('value', 0)))
PY
  done
done
