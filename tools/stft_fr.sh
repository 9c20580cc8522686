#!/bin/bash
# developer A/B (GPU box): frames per STFT workgroup (-DSE_STFT_FR): head workload linear201 (no mel plane) and the enhance pass
cd "$(dirname "$0")/.."
for fr in ${SE_FR_LIST:-30 20 30 20}; do
  SE_AMD_EXTRA_DEFINES="-DSE_STFT_FR=$fr $SE_FR_EXTRA" python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1 || { echo "build failed $fr"; continue; }
  echo -n "FR=$fr head: "; python3 bench.py --workload head --head-feat linear201 --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'stft', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
  echo -n "FR=$fr head mel120: "; python3 bench.py --workload head --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'stft', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
  echo -n "FR=$fr enhance: "; python3 bench.py --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'stft', round(d['roofline_other_kernels']['stft_kernel']['avg_launch_ms']*1e3,1), 'us')"
done
python3 speech-enhancement-by-s3prl_amd/build.py > /dev/null 2>&1
