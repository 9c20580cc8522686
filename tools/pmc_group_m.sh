#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / time of the persistent QKV and FFN1 GEMMs for several tile-group heights (SE_AMD_GEMM_GROUPM): does the L2-level
# over-fetch of the A panels / weight matrix move with the tile order, and does the launch time follow it?  
#   tools/pmc_group_m.sh
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_group_m
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for gm in 1 2 4 8 16 126; do
  export SE_AMD_GEMM_GROUPM=$gm
  python3 "$root/tools/bench_kernels.py" qkv > "$out/time_$gm.txt" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/g${gm}_$c" -o x -- python3 "$root/tools/bench_kernels.py" qkv > "$out/g${gm}_$c.log" 2>&1
  done
  python3 "$root/tools/pmc_summary.py" "$out/group_m_$gm.json" $(find "$out/g${gm}_FETCH_SIZE" "$out/g${gm}_WRITE_SIZE" -name '*counter_collection.csv') > /dev/null
  rm -rf "$out/g${gm}_FETCH_SIZE" "$out/g${gm}_WRITE_SIZE"
  echo "group_m=$gm"; grep -v amdgpu.ids "$out/time_$gm.txt"
  python3 - "$out/group_m_$gm.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if 'gemm6' in k:
        print('   ', k[:48], 'fetch x2 MB', round(2 * v.get('FETCH_SIZE_KB_mean', 0) / 1024, 1), 'write MB', round(v.get('WRITE_SIZE_KB_mean', 0) / 1024, 1))
PY
done
