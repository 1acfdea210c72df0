#!/usr/bin/env python3
"""Per-kernel difference of two rocprofv3 kernel_stats.csv runs, divided by the number of extra calls between them
(usage: kdiff.py <dir_few> <dir_many> <extra calls> [title]): the kernels ONE call launches, in calls, ms and average us."""
import csv
import glob
import os
import sys


def load(d):
    f = max(glob.glob(d + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
    out = {}
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('he355::', '').replace('ks_fold::', 'F:').replace('ks_shoup::', 'S:').replace('void ', '').split('(')[0]
        c, t = out.get(n, (0, 0.0))
        out[n] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
    return out


a, b, extra = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
title = sys.argv[4] if len(sys.argv) > 4 else ""
rows = []
for n, (cb, tb) in b.items():
    ca, ta = a.get(n, (0, 0.0))
    dc, dt = (cb - ca) / extra, (tb - ta) / extra / 1e6
    if dc > 0.01:
        rows.append((dt, n, dc))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# {title}: kernels of one operate() call")
for dt, n, dc in rows:
    print(f"  {n[:46]:46s} calls {dc:7.1f}  ms {dt:8.3f}  avg_us {dt / dc * 1e3:8.1f}  {100 * dt / tot:5.1f} %")
print(f"  sum of kernels per operate(): {tot:.3f} ms in {sum(r[2] for r in rows):.0f} launches")
