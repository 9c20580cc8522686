set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python3 bench.py --workload head --no-cpu-baseline --no-host-fed > gpurun_out/r03head_bench.json 2> gpurun_out/r03head_bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03head_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('batch_12',{}).get('value'))
for k,v in d['roofline_other_kernels'].items():
    if isinstance(v,dict): print(k, round(v['frac'],3), round(v['avg_launch_ms']*1e3,1))
print(d['pass_hbm']['frac_sum_of_kernels'])
PY
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-host-fed --no-extras > gpurun_out/r03glue_bench.json 2> gpurun_out/r03glue_bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03glue_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
tools/prof_stats.sh r03glue2 --steps 10 --warmup 2 2>&1 | grep -v "gemm6p\|gemm7" | tail -20
