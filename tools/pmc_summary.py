#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv (+ kernel_trace.csv durations) per kernel name."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("he355::", "").replace("(anonymous namespace)::", "")
    build = "F:" if "ks_fold::" in name else "S:" if "ks_shoup::" in name else ""  # which build of the device code (fold / Shoup form of the u64 engine)
    name = name.replace("ks_fold::", "").replace("ks_shoup::", "")
    m = re.search(r"(k_\w+)(<[^(]*>)?", name)
    return build + (m.group(1) + (m.group(2) or "")).replace(" ", "").replace(",", ";") if m else name[:40]


def main(d):
    import glob
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), short(r["Kernel_Name"]))
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(cc)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    tdur = defaultdict(float)
    for did, (ns, k) in dur.items():
        tdur[k] += ns
    names = sorted({c for k in agg for c in agg[k]})
    print("kernel,calls,total_us," + ",".join(names))
    for k in sorted(agg, key=lambda k: -tdur[k]):
        print(f"{k},{len(calls[k])},{tdur[k]/1e3:.1f}," + ",".join(f"{agg[k][c]:.4g}" for c in names))


if __name__ == "__main__":
    main(sys.argv[1])
