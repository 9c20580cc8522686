#!/bin/bash
# Round-end evidence run on the GPU box (everything under gpurun_out/<tag>/): full GPU test-suite, the profile set of
# tools/collect_profiles.sh, the side workloads' bench lines, per-kernel SQ / matrix-pipe counter passes, the micro-benchmarks.
#   tools/final_evidence.sh r02b [part]      part 1: tests + the enhance / head profile sets, part 2: the rest (default: both; one gpurun call
#                                            is limited to 20 minutes)
set -e
tag=$1
part=${2:-0}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
if [ "$part" != 2 ]; then
python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1 || { tail -30 "$out/pytest_gpu.log"; exit 1; }
tail -3 "$out/pytest_gpu.log"
cp gpurun_out/parity_measured.txt "$out/${tag}_parity_measured.txt" 2>/dev/null || true
tools/collect_profiles.sh $tag
tools/collect_profiles.sh $tag _head_mel120 --workload head --head-feat mel120
tools/collect_profiles.sh $tag _head_linear201 --workload head --head-feat linear201
fi
[ "$part" = 1 ] && exit 0
if [ "$part" != 3 ]; then      # part 3: only what follows the side-workload lines (re-run after a failure further down)
tools/collect_profiles.sh $tag _finetune --workload finetune
cd "$root"
python3 bench.py --workload lstm --no-cpu-baseline > "$out/${tag}_lstm_bench.json" 2> "$out/lstm.err"
python3 tools/bench_kernels.py mhsa_train > "$out/${tag}_mhsa_train.txt" 2>&1
python3 tools/bench_kernels.py mhsa_peaked > "$out/${tag}_mhsa_peaked.txt" 2>&1
fi
python3 tools/bench_kernels.py all > "$out/${tag}_bench_kernels.txt" 2>&1
python3 tools/bench_kernels.py hbm >> "$out/${tag}_bench_kernels.txt" 2>&1
SE_AMD_GEMM_SMALL_M=0 python3 tools/bench_kernels.py square >> "$out/${tag}_bench_kernels.txt" 2>&1
python3 tools/bench_kernels.py ksweep >> "$out/${tag}_bench_kernels.txt" 2>&1
[ -x tools/micro/valu_rate ] || /opt/rocm/bin/hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -o tools/micro/valu_rate tools/micro/valu_rate.hip
tools/micro/valu_rate > "$out/${tag}_micro_valu_rate.txt" 2>&1
for k in gemm gemmln mhsa stft; do
  tools/pmc_sq.sh ${tag}_$k tools/bench_kernels.py $k > "$out/pmc_sq_$k.log" 2>&1
  cp "$root/gpurun_out/pmc_${tag}_$k/summary.json" "$out/${tag}_pmc_sq_$k.json"
done
ls "$out"
