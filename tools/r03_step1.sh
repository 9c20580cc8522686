#!/bin/bash
# round 3, step 1: the new bench.py (phase A / phase B), the head workload, the 2-rank fine-tune rehearsal that deadlocked in round 2
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r03s1
mkdir -p "$out"
cd "$root"
timeout -k 10 300 python3 -m pytest tests/test_gpu_encoder_pipeline.py -x -q -m gpu -k "head_enhance" > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
timeout -k 10 300 python3 bench.py --workload head > "$out/head_mel120.json" 2> "$out/head_mel120.err" || { tail -20 "$out/head_mel120.err"; exit 1; }
tail -c 2500 "$out/head_mel120.json"
timeout -k 10 300 python3 bench.py --workload head --head-feat linear201 > "$out/head_linear201.json" 2> "$out/head_linear201.err" || { tail -20 "$out/head_linear201.err"; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 2 --workload finetune --one-device --backend gloo --steps 5 --warmup 2 > "$out/b_2rank_ft.json" 2> "$out/b_2rank_ft.err" || { tail -20 "$out/b_2rank_ft.err"; exit 1; }
tail -c 1200 "$out/b_2rank_ft.json"
timeout -k 10 300 python3 bench.py --gpus 2 --one-device --backend gloo --steps 5 --warmup 2 > "$out/b_2rank.json" 2> "$out/b_2rank.err" || { tail -20 "$out/b_2rank.err"; exit 1; }
timeout -k 10 300 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
tail -c 1500 "$out/bench.json"
