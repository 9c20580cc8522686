#!/bin/bash
# whole-pass A/B of gemm6q: SE_AMD_GEMM6Q = 0 (gemm6p), 1 (identity GEMMs on gemm6q), 3 (identity + GELU) with the product library and tagged ones
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05e
mkdir -p "$out"
cd "$root"
f="$out/r05_gemm6q_pass_ab_$(echo $@ | tr ' ' '_').txt"
: > "$f"
en() { timeout -k 10 300 python3 bench.py --steps 40 --no-extras --no-cpu-baseline --no-host-fed --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do
  line="pass ms:"
  for lib in so $@; do
    L=$root/speech-enhancement-by-s3prl_amd/libse_amd.$lib; [ "$lib" = so ] || L=$L.so
    for q in 0 1 3; do line="$line | $lib q=$q $(SE_AMD_LIB=$L SE_AMD_GEMM6Q=$q en)"; done
  done
  echo "$line" | tee -a "$f"
done
