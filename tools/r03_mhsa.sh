set -e
cd $GRAFT_REPO_ROOT
for d in 3 4; do echo "tests dma $d"; SE_AMD_MHSA_DMA=$d timeout -k 10 600 python3 -m pytest tests/test_gpu_encoder_blocks.py -x -q -m gpu -k mhsa 2>&1 | tail -1; done
for d in 0 1 3 4 0 1 3 4; do echo -n "dma $d: "; SE_AMD_MHSA_DMA=$d timeout -k 5 120 python3 tools/bench_kernels.py mhsa 2>&1 | grep prescaled | cut -c1-120; done
