#!/bin/bash
# whole-pass sweep of the library's runtime switches on one box (each point: bench.py, 40 steps), two rounds
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05k
mkdir -p "$out"
cd "$root"
f="$out/r05_knob_sweep.txt"
: > "$f"
en() { timeout -k 10 300 python3 bench.py --steps 40 --no-extras --no-cpu-baseline --no-host-fed --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for rep in 1 2; do
  for kv in "X=0" "SE_AMD_GEMM7_INM=0" "SE_AMD_GEMM6P_INM=0" "SE_AMD_GEMM6P_INM=1" "SE_AMD_GEMM_GROUPM=2" "SE_AMD_GEMM_GROUPM=8" "SE_AMD_GEMM6P_LATE=0" "SE_AMD_GEMM6P_LATE=1" "SE_AMD_GEMM6P_LATE=3" "SE_AMD_MHSA_PIPE=0" "SE_AMD_GEMM4_STAGGER=2" "SE_AMD_STFT_SMALL=0" "SE_AMD_MHSA_SPEC=0" "X=1"; do
    echo "$kv: $(env $kv bash -c "$(declare -f en); en")" | tee -a "$f"
  done
done
