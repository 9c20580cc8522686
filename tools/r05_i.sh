#!/bin/bash
# STFT frames-per-workgroup sweep of the small kernel (pass B = 8 items per frame: 8 / 16 / 32 frames fill whole waves), stand-alone and in the pass
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05i
mkdir -p "$out"
cd "$root"
f="$out/r05_stft_fr_sweep.txt"
: > "$f"
en() { timeout -k 10 300 python3 bench.py --steps 40 --no-extras --no-cpu-baseline --no-host-fed 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'stft us', round(d['roofline_kernels']['stft_kernel']['avg_launch_ms']*1e3,1) if 'roofline_kernels' in d else '')"; }
for rep in 1 2; do for lib in so fr8.so fr16.so fr32.so; do
echo "== libse_amd.$lib" | tee -a "$f"
SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/libse_amd.$lib timeout -k 10 200 python3 tools/bench_kernels.py stft 2>&1 | grep "phasor 2ch\|phasor 1ch" | cut -c1-120 | tee -a "$f"
echo "pass: $(SE_AMD_LIB=$root/speech-enhancement-by-s3prl_amd/libse_amd.$lib en)" | tee -a "$f"
done; done
