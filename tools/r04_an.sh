#!/bin/bash
# end-of-round re-check of every default that was decided by a small margin: interleaved pairs on ONE box
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04an
mkdir -p "$out"
cd "$root"
f="$out/r04_decisions_recheck.txt"
: > "$f"
ft() { timeout -k 10 300 python3 bench.py --workload finetune --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
en() { timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do
  echo "fine-tune step ms: default $(ft) | SE_AMD_GEMM7_PLAIN=0 $(SE_AMD_GEMM7_PLAIN=0 ft) | SE_AMD_GEMM6_DUAL=0 $(SE_AMD_GEMM6_DUAL=0 ft) | SE_AMD_WGRAD_STAG=0 $(SE_AMD_WGRAD_STAG=0 ft)" | tee -a "$f"
done
for rep in 1 2 3; do
  echo "enhance pass ms: default $(en) | SE_AMD_MHSA_PIPE=0 (round-3 attention kernel) $(SE_AMD_MHSA_PIPE=0 en) | SE_AMD_MHSA_PIPE=11 (persistent) $(SE_AMD_MHSA_PIPE=11 en)" | tee -a "$f"
done
