#!/bin/bash
# round 5: (1) the new / changed GPU tests, (2) the four-rank one-device gloo rehearsal of the fine-tune step ONCE with the host-staged buckets
# (all seven in flight as device -> host copies; gloo sees host tensors only)
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05g
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist_finetune.py tests/test_gpu_checkpoint_feeder.py tests/test_gpu_preprocessor.py tests/test_gpu_encoder_blocks.py tests/test_gpu_graph.py -x -q -k "dist or checkpoint or inference_mode or lazy_phase or dual_gelu or large_m or x3 or gemm_vs_torch" 2>&1 | tail -8 | tee "$out/tests.txt"
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
SE_BENCH_WATCHDOG=90 timeout -k 10 150 python3 bench.py --gpus 4 --workload finetune --one-device --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-extras > "$out/gloo4_bench.json" 2> "$out/gloo4_stderr.txt"
rc=$?
echo "four-rank gloo rehearsal rc=$rc $(tail -c 400 "$out/gloo4_bench.json")" | tee "$out/gloo4_summary.txt"
if [ $rc -ne 0 ]; then grep -n "watchdog\|most recent call first\|File \"" "$out/gloo4_stderr.txt" | head -80 | tee -a "$out/gloo4_summary.txt"; fi
exit 0
