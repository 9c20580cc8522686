#!/usr/bin/env python
"""Merge rocprofv3 --pmc counter_collection CSVs (one pass per counter) into {kernel: {COUNTER_KB_mean, launches}}.
    python tools/pmc_summary.py out.json pass1_counter_collection.csv pass2_counter_collection.csv ..."""
import collections
import csv
import json
import re
import sys

out_path, paths = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in paths:
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).strip()
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for name, counters in acc.items():
    d = {}
    for c, vals in counters.items():
        d[c + '_KB_mean'] = round(sum(vals) / len(vals), 1)      # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
        d['launches'] = len(vals)
    res[name] = d
json.dump(res, open(out_path, 'w'), indent=1)
print('wrote', out_path, len(res), 'kernels')
