#!/usr/bin/env python
"""Merge rocprofv3 --pmc counter_collection CSVs into per-kernel means and the derived matrix-pipe figures:
    mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES-equivalent): here reported against SQ_BUSY_CYCLES x #SIMD share
    python tools/pmc_sq_summary.py out.json *.csv"""
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
        name = re.sub(r'^void ', '', name)
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for name, counters in acc.items():
    d = {c: sum(v) / len(v) for c, v in counters.items()}
    d['launches'] = max(len(v) for v in counters.values())
    wc = d.get('SQ_WAVE_CYCLES')
    if wc:
        for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS'):
            if k in d:
                d[k + '_per_wave_cycle'] = round(d[k] / wc, 4)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'SQ_BUSY_CYCLES' in d and d['SQ_BUSY_CYCLES']:
        # SQ_BUSY_CYCLES is summed over the shader engines' SQs; MFMA busy cycles over SIMDs: report the raw ratio and let the reader
        # compare kernels (a kernel with every SIMD's matrix pipe always busy gives the same constant for all)
        d['mfma_busy_over_sq_busy'] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES'], 4)
    if 'SQ_LDS_BANK_CONFLICT' in d and d.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict_frac'] = round(d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE'], 4)
    if 'TCC_HIT_sum' in d and (d['TCC_HIT_sum'] + d.get('TCC_MISS_sum', 0)):
        d['l2_hit_rate'] = round(d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum']), 4)
    res[name] = {k: (round(v, 1) if isinstance(v, float) and abs(v) > 10 else v) for k, v in d.items()}
json.dump(res, open(out_path, 'w'), indent=1, sort_keys=True)
for name, d in sorted(res.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:14]:
    keys = ('launches', 'mfma_busy_over_sq_busy', 'SQ_WAIT_ANY_per_wave_cycle', 'SQ_WAIT_INST_ANY_per_wave_cycle', 'SQ_ACTIVE_INST_ANY_per_wave_cycle',
            'SQ_ACTIVE_INST_VALU_per_wave_cycle', 'SQ_ACTIVE_INST_LDS_per_wave_cycle', 'lds_bank_conflict_frac', 'l2_hit_rate')
    print(name[:70], {k.replace('_per_wave_cycle', '').replace('SQ_', ''): d[k] for k in keys if k in d})
