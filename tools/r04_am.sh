#!/bin/bash
# batch sweep of the evaluate()-style pass (bf16 path): row-complete fused projections (default) vs separate GEMM + LayerNorm launches (SE_AMD_FUSED_LN=0)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04am
mkdir -p "$out"
cd "$root"
: > "$out/r04_batch_sweep.txt"
for b in 1 2 4 8 12 16 24 32; do for f in 1 0; do
  ms=$(SE_AMD_FUSED_LN=$f timeout -k 10 300 python3 bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value']))")
  echo "batch $b SE_AMD_FUSED_LN=$f: $ms (ms per pass, utt/s)" | tee -a "$out/r04_batch_sweep.txt"
done; done
