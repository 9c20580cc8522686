#!/bin/bash
# four ranks on one GPU over gloo (device tensors), fine-tune step: how many bucket all-reduces may be in flight?  (round 3: 7 hung, 1 passes)
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04j
mkdir -p "$out"
cd "$root"
for cap in 1 2 3 7; do
  echo "=== SE_DP_GLOO_INFLIGHT=$cap"
  SE_DP_GLOO_INFLIGHT=$cap SE_BENCH_WATCHDOG=90 timeout -k 10 150 python3 bench.py --gpus 4 --workload finetune --one-device --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-extras > "$out/cap$cap.json" 2> "$out/cap$cap.err"
  rc=$?
  echo "cap $cap rc=$rc $(tail -c 300 "$out/cap$cap.json" | head -c 300)"
  if [ $rc -ne 0 ]; then grep -n "watchdog\|most recent call first\|File \"" "$out/cap$cap.err" | head -60; fi
done 2>&1 | tee "$out/r04_gloo_inflight.txt"
exit 0
