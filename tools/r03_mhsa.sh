set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_encoder_blocks.py -x -q -m gpu -k mhsa 2>&1 | tail -2
for p in 0 3 0 3 0 3; do echo -n "pipe $p: "; SE_AMD_MHSA_PIPE=$p timeout -k 5 120 python3 tools/bench_kernels.py mhsa 2>&1 | grep prescaled | cut -c1-120; done
