#!/bin/bash
# Per-kernel SQ / matrix-pipe / LDS counters (separate rocprofv3 --pmc passes, 8 SQ slots each) of a command, merged by tools/pmc_sq_summary.py.
#   tools/pmc_sq.sh <tag> <program args...>      e.g.  tools/pmc_sq.sh kern tools/bench_kernels.py gemm
# The program runs directly after `--` (the profiler's preload initialises the GPU: no env / bash -c / launcher hop in between).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -o p$i -- python3 "$root/$1" "${@:2}" > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/p$i.log"; }
done
python3 "$root/tools/pmc_sq_summary.py" "$out/summary.json" $(find "$out" -name '*counter_collection.csv') | tail -40
# keep only the summary and the logs (the raw CSVs are large)
find "$out" -name '*.csv' -delete
