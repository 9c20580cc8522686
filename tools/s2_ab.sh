#!/bin/bash
# whole-pass sweep of SE_AMD_GEMM6P_LATE (start delay, in ~4 us units, of the persistent GEMM's workgroups whose tile list is one shorter) on one box, alternated
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 40 --warmup 5 --no-roofline --no-cpu-baseline --no-host-fed --no-extras "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1))"; }
for r in 1 2 3; do
  for l in 2 3 4 5 6 8; do echo -n "LATE=$l: "; SE_AMD_GEMM6P_LATE=$l run; done
done
