for s in 1 2 1 2; do echo SCHED=$s; SE_AMD_GEMM3_SCHED=$s timeout -k 10 200 python tools/bench_kernels.py gemm 2>&1 | grep -E "N=2304|N=3072"; done
