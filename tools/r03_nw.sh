set -e
cd $GRAFT_REPO_ROOT
SE_AMD_MHSA_NW=8 timeout -k 10 600 python3 -m pytest tests/test_gpu_encoder_blocks.py -x -q -m gpu -k mhsa 2>&1 | tail -2
for p in 4 8 4 8 4 8; do echo -n "nw $p: "; SE_AMD_MHSA_NW=$p timeout -k 5 120 python3 tools/bench_kernels.py mhsa 2>&1 | grep prescaled | cut -c1-120; done
