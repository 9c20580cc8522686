#!/bin/bash
# developer A/B (GPU box): LDS footprint / occupancy of the linear-head kernel (-DSE_HEAD_HK=<K chunk> -DSE_HEAD_HALF=<rows per epilogue pass>)
cd "$(dirname "$0")/.."
for cfg in "40 16" "20 8" "40 8" "20 16" "40 16"; do
  set -- $cfg
  SE_AMD_EXTRA_DEFINES="-DSE_HEAD_HK=$1 -DSE_HEAD_HALF=$2" python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1 || { echo "build failed $cfg"; continue; }
  for feat in mel120 linear201; do
  echo -n "HK=$1 HALF=$2 $feat: "; python3 bench.py --workload head --head-feat $feat --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'head', round(d['roofline_other_kernels']['head_kernel']['avg_launch_ms']*1e3,1), 'us')"
  done
done
python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1
