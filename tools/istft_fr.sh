#!/bin/bash
# developer A/B (GPU box): frames per iSTFT workgroup (-DSE_ISTFT_FR): head workload (B = 256) and the enhance pass (B = 32)
cd "$(dirname "$0")/.."
for fr in ${SE_FR_LIST:-30 20 14 10 30}; do
  SE_AMD_EXTRA_DEFINES="-DSE_ISTFT_FR=$fr" python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1 || { echo "build failed $fr"; continue; }
  echo -n "FR=$fr head: "; python3 bench.py --workload head --head-feat linear201 --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'istft', round(d['roofline_other_kernels']['istft_kernel']['avg_launch_ms']*1e3,1), 'us')"
  echo -n "FR=$fr enhance: "; python3 bench.py --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'istft', round(d['roofline_other_kernels']['istft_kernel']['avg_launch_ms']*1e3,1), 'us')"
done
python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1
