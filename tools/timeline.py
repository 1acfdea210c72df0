#!/usr/bin/env python3
"""Print the kernel timeline of a rocprofv3 kernel trace (usage: timeline.py <dir> [first_n]): queue, start/end offsets in ms."""
import csv
import glob
import os
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = max(glob.glob(d + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'fill_uniform' not in r['Kernel_Name'] and 'key_to_engine' not in r['Kernel_Name'] and 'key_quot' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for r in rows[skip:skip + first]:
    n = r['Kernel_Name'].replace('he355::(anonymous namespace)::', '').replace('he355::', '').split('(')[0].replace('void ', '')
    s, e = (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6
    print(f"q{r['Queue_Id']:>3s} {s:9.3f} {e:9.3f} {e - s:8.3f}  {n[:48]}")
