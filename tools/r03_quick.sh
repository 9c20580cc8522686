set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
tools/prof_stats.sh r03glue3 --steps 10 --warmup 2 2>&1 | grep -v "gemm6p\|gemm7" | head -16
