# developer sweep: evaluate()-pass latency vs batch for the GEMM dispatch variants (SE_AMD_GEMM) with / without the fused GEMM+LN kernel
for b in 1 2 4 8 16 32; do
  for cfg in "5 1" "5 0" "4 0" "4 1"; do
    set -- $cfg
    ms=$(SE_AMD_GEMM=$1 SE_AMD_FUSED_LN=$2 timeout -k 10 300 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
    echo "batch $b GEMM=$1 FUSED_LN=$2: $ms ms"
  done
done
