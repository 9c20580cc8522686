#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel totals divided by the number of steps."""
import csv
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total GPU-busy ms / step: {tot / 1e6 / steps:.3f}   ({len(rows)} kernels)')
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} calls/step={float(r['Calls']) / steps:8.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / steps:8.3f} "
          f"avg_us={float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):5.1f}%")
