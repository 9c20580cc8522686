#!/bin/bash
# Whole-pass A/B on ONE box, settings alternated inside the call (the pool's boxes differ by 4-5 %: only same-call comparisons mean anything).
#   tools/pass_ab.sh                     headline pass: default | two half batches on two streams | SE_AMD_GEMM6P_LATE 3, 4
#   tools/pass_ab.sh head                configs[3] pass: default | 20-frame STFT build | hipGraph replay | linear-spectrogram features
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
what=${1:-enhance}
run() { python3 bench.py --steps 40 --warmup 5 --no-roofline --no-cpu-baseline --no-host-fed --no-extras "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value'],1))"; }
for r in 1 2 3; do
  if [ "$what" = head ]; then
    echo -n "default:      "; run --workload head
    echo -n "STFT_SMALL=0: "; SE_AMD_STFT_SMALL=0 run --workload head
    echo -n "graph:        "; run --workload head --graph
    echo -n "linear201:    "; run --workload head --head-feat linear201
  else
    echo -n "default:   "; run
    echo -n "streams 2: "; run --streams 2
    for l in 3 4; do echo -n "LATE=$l:    "; SE_AMD_GEMM6P_LATE=$l run; done
  fi
done
