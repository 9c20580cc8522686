#!/bin/bash
# developer tool (GPU box): mhsa3.hip with parts compiled out (-DSE_MHSA3_ABL=<mask>; results are wrong, only time matters)
cd "$(dirname "$0")/.."
for m in ${SE_ABL_LIST:-0 1 2 3 4 8 16 12 28 31}; do
  SE_AMD_EXTRA_DEFINES=-DSE_MHSA3_ABL=$m python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1 || { echo "build failed for $m"; continue; }
  echo -n "abl3 $m: "; SE_AMD_MHSA_PIPE=3 timeout -k 5 120 python3 tools/bench_kernels.py mhsa 2>&1 | grep prescaled | cut -c1-70
done
python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1
