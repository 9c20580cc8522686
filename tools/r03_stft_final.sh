set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests/test_gpu_preprocessor.py tests/test_gpu_encoder_pipeline.py tests/test_gpu_fullsize_properties.py tests/test_gpu_dropin_sequence.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1 0 1; do
  echo -n "small=$v head linear201: "; SE_AMD_STFT_SMALL=$v python3 bench.py --workload head --head-feat linear201 --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'stft', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
done
echo -n "head mel120: "; python3 bench.py --workload head --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'stft', round(d['roofline']['avg_launch_ms']*1e3,1), 'us')"
echo -n "enhance: "; python3 bench.py --no-cpu-baseline --no-host-fed --no-extras 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3), 'stft', round(d['roofline_other_kernels']['stft_kernel']['avg_launch_ms']*1e3,1), 'us')"
