#!/bin/bash
# developer tool (GPU box): wave priority around the MFMA clusters of the inference MHSA forward (-DSE_MHSA_PRIO=<mask>)
cd "$(dirname "$0")/.."
for m in ${SE_PRIO_LIST:-0 3 1 2 0 3}; do
  SE_AMD_EXTRA_DEFINES="-DSE_MHSA_PRIO=$m" python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1 || { echo "build failed for $m"; continue; }
  echo -n "prio $m: "; timeout -k 5 120 python3 tools/bench_kernels.py mhsa 2>&1 | grep prescaled | cut -c1-70
done
python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1
