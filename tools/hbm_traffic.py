#!/usr/bin/env python3
"""HBM bytes per op from the two PMC passes of tools/profile_round.sh (usage: hbm_traffic.py <fetch_dir> <write_dir> <ops> <out.json>).
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of a wide coalesced read, so it is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Setup kernels (fill, key preparation, copies) are left out."""
import csv
import glob
import json
import sys
from collections import defaultdict

from pmc_summary import short

SETUP = ("k_fill_uniform", "k_key_to_engine", "k_key_quotients", "__amd_rocclr")


def collect(d, counter):
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    val, us = defaultdict(float), defaultdict(float)
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == counter:
            val[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    for r in csv.DictReader(open(kt)):
        us[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return val, us


def main(fetch_dir, write_dir, ops, out):
    ops = float(ops)
    rd, us = collect(fetch_dir, "FETCH_SIZE")
    wr, _ = collect(write_dir, "WRITE_SIZE")
    per, total = {}, 0.0
    for k in sorted(us, key=lambda k: -us[k]):
        if k.startswith(SETUP) or any(s in k for s in SETUP):
            continue
        r, w = 2.0 * rd.get(k, 0.0) * 1024 / ops, wr.get(k, 0.0) * 1024 / ops
        total += r + w
        per[k] = {"read": round(r / 2**20, 2), "write": round(w / 2**20, 2), "us": round(us[k] / ops, 2)}
    json.dump({"hbm_bytes_per_op": round(total, 0),
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one bench step (%d ops, single stream); FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); counter unit KB" % int(ops),
               "per_kernel_MiB_per_op": per}, open(out, "w"), indent=1)
    print("hbm MiB/op: %.1f" % (total / 2**20))


if __name__ == "__main__":
    main(*sys.argv[1:5])
