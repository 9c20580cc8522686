#!/bin/bash
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r04ad
mkdir -p "$out"
cd "$root"
: > "$out/r04_nt_stores.txt"
for rep in 1 2 3; do for lib in libse_amd.so libse_amd.nt.so; do
echo "== $lib" | tee -a "$out/r04_nt_stores.txt"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 300 python3 tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids | grep "N=2304\|N=3072" | tee -a "$out/r04_nt_stores.txt"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/$lib timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline > "$out/b.json" 2> "$out/err" || { tail -20 "$out/err"; exit 1; }
python3 -c "
import json; d = json.loads(open('$out/b.json').read().strip().splitlines()[-1]); print('enhance', d['value'], d['unit'], d['ms_per_step'], 'ms')" | tee -a "$out/r04_nt_stores.txt"
done; done
