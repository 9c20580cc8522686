#!/bin/bash
# developer tool (GPU box): the inference MHSA forward with parts of the tile body compiled out (-DSE_MHSA_ABL=<mask>; results are wrong, only time matters)
cd "$(dirname "$0")/.."
for m in ${SE_ABL_LIST:-0 1 2 3 4 8 16 24 32 28 60 63}; do
  SE_AMD_EXTRA_DEFINES=-DSE_MHSA_ABL=$m python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1 || { echo "build failed for $m"; continue; }
  echo -n "abl $m: "; timeout -k 5 120 python3 tools/bench_kernels.py mhsa 2>&1 | grep prescaled | cut -c1-70
done
python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1
