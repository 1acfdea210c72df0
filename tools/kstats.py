#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as per-step milliseconds (usage: kstats.py <dir> <steps incl. warm-up>)."""
import csv
import glob
import os
import sys

d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = max(glob.glob(d + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('he355::', '').replace('ks_fold::', 'F:').replace('ks_shoup::', 'S:').replace('void ', '').split('(')[0]
    ms = float(r['TotalDurationNs']) / 1e6 / steps
    if 'fill_uniform' in n or 'key_to_engine' in n or 'copyBuffer' in n:
        continue
    tot += ms
    if float(r['Percentage']) >= 0.5:
        print(f"  {n[:44]:44s} calls {r['Calls']:>4s}  ms/step {ms:8.2f}  avg_us {float(r['AverageNs']) / 1e3:9.1f}")
print(f"  sum of kernels per step: {tot:.2f} ms")
