#!/bin/bash
# round-4 call A: lds_rate micro, MHSA stamps at 1/2/3 waves per SIMD, default bench line + kernel micro-benchmarks of today's box
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04a
mkdir -p "$out"
cd "$root"
tools/micro/lds_rate > "$out/r04_micro_lds_rate.txt" 2>&1
cat "$out/r04_micro_lds_rate.txt"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/libse_amd.stamps.so timeout -k 10 200 python3 tools/mhsa_stamps.py > "$out/r04_mhsa_stamps.txt" 2>&1
cat "$out/r04_mhsa_stamps.txt"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-host-fed > "$out/r04a_bench.json" 2> "$out/bench.err"
tail -c 1200 "$out/r04a_bench.json"
timeout -k 10 300 python3 tools/bench_kernels.py all > "$out/r04a_bench_kernels.txt" 2>&1
cat "$out/r04a_bench_kernels.txt"
