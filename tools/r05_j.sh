#!/bin/bash
# fine-tune step: this tree against the tree of commit c43a369 (worktree _old/, before the dist.py / prune changes), alternated on one box
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/r05j
mkdir -p "$out"
ft() { (cd "$1" && timeout -k 10 300 python3 bench.py --workload finetune --no-extras --no-cpu-baseline --no-host-fed --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"); }
for rep in 1 2 3; do echo "fine-tune ms: new $(ft $root) | old $(ft $root/_old)"; done | tee "$out/r05_finetune_old_new.txt"
