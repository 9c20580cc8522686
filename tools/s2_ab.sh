#!/bin/bash
# configs[3] pass, one box, alternated: default | the 20-frame STFT build (SE_AMD_STFT_SMALL=0)
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --workload head --steps 40 --warmup 5 --no-roofline --no-cpu-baseline --no-host-fed --no-extras "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1))"; }
python -m pytest tests/test_gpu_fused_head_pass.py -x -q 2>&1 | tail -2
for r in 1 2 3; do
  echo -n "default:      "; run
  echo -n "STFT_SMALL=0: "; SE_AMD_STFT_SMALL=0 run
  echo -n "graph:        "; run --graph
  echo -n "linear201:    "; run --head-feat linear201
done
