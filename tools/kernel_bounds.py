#!/usr/bin/env python3
"""Per-kernel binding resource from three rocprofv3 PMC passes of ONE bench step (usage: kernel_bounds.py <fetch_dir> <write_dir> <sq_dir> <ops> [--json out]).

  HBM   : (2 x FETCH_SIZE + WRITE_SIZE) KB / kernel time, against 8 TB/s (spec) -- FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
  VALU  : SQ_INSTS_VALU wave-instructions x 4 cycles (a wave64 instruction occupies its 16-lane SIMD for 4 cycles) / (1024 SIMDs x kernel cycles),
          kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs of the same pass (so the sustained clock is the one the kernel ran at)
  bound : the larger of the two fractions names the binding resource; below 0.35 for both the kernel is latency / occupancy bound ("neither").
"""
import csv
import glob
import json
import sys
from collections import defaultdict

from pmc_summary import short

SETUP = ("k_fill_uniform", "k_key_to_engine", "k_key_quotients", "k_key_fold", "__amd_rocclr", "k_key_scaled_copy")
HBM_PEAK, N_SIMD, N_XCD = 8.0e12, 1024, 8


def collect(d):
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    val, us, calls = defaultdict(lambda: defaultdict(float)), defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(cc)):
        val[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for r in csv.DictReader(open(kt)):
        k = short(r["Kernel_Name"])
        us[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        calls[k] += 1
    return val, us, calls


def table(fetch_dir, write_dir, sq_dir, ops):
    rd, us_f, _ = collect(fetch_dir)
    wr, us_w, _ = collect(write_dir)
    sq, us_s, calls = collect(sq_dir)
    rows = []
    for k in sorted(us_s, key=lambda k: -us_s[k]):
        if any(s in k for s in SETUP):
            continue
        rbytes = 2.0 * rd[k].get("FETCH_SIZE", 0.0) * 1024
        wbytes = wr[k].get("WRITE_SIZE", 0.0) * 1024
        t_mem = (us_f.get(k, 0) + us_w.get(k, 0)) / 2 * 1e-6 or us_s[k] * 1e-6  # the byte passes' own kernel time
        cyc = sq[k].get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        insts = sq[k].get("SQ_INSTS_VALU", 0.0)
        hbm_frac = (rbytes + wbytes) / t_mem / HBM_PEAK if t_mem else 0.0
        valu_frac = insts * 4 / (N_SIMD * cyc) if cyc else 0.0
        mhz = cyc / us_s[k] if us_s[k] else 0.0
        bound = "VALU issue" if valu_frac >= hbm_frac and valu_frac >= 0.35 else "HBM" if hbm_frac > valu_frac and hbm_frac >= 0.35 else "neither (latency / occupancy)"
        rows.append({"kernel": k, "calls": calls[k], "ms_per_step": round(us_s[k] / 1e3, 3), "read_MiB_per_op": round(rbytes / ops / 2**20, 3),
                     "write_MiB_per_op": round(wbytes / ops / 2**20, 3), "hbm_TBps": round((rbytes + wbytes) / t_mem / 1e12, 2) if t_mem else 0.0,
                     "hbm_frac_of_8TBps": round(hbm_frac, 3), "valu_wave_instr": insts, "valu_issue_frac": round(valu_frac, 3),
                     "lds_wave_instr": sq[k].get("SQ_INSTS_LDS", 0.0), "sustained_mhz": round(mhz), "bound": bound})
    return rows


def main():
    fetch_dir, write_dir, sq_dir, ops = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    rows = table(fetch_dir, write_dir, sq_dir, ops)
    tot_ms = sum(r["ms_per_step"] for r in rows)
    tot_i = sum(r["valu_wave_instr"] for r in rows)
    tot_b = sum((r["read_MiB_per_op"] + r["write_MiB_per_op"]) for r in rows)
    if "--json" in sys.argv:
        long_k = [r for r in rows if r["ms_per_step"] >= 1.0] or rows
        json.dump({"ops_per_step": ops, "ms_per_step_single_stream_under_counters": round(tot_ms, 3), "hbm_bytes_per_op": round(tot_b * 2**20),
                   "valu_wave_instr_per_op": round(tot_i / ops), "valu_wave_instr_per_step": tot_i,
                   "sustained_mhz_time_weighted": round(sum(r["sustained_mhz"] * r["ms_per_step"] for r in long_k) / sum(r["ms_per_step"] for r in long_k)),
                   "method": "three rocprofv3 PMC passes (FETCH_SIZE; WRITE_SIZE; SQ_* + GRBM_GUI_ACTIVE), each its own run of ONE bench step with --kernel-trace only, "
                             "single stream; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md); VALU issue fraction = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x "
                             "GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / kernel time (reads low on kernels whose XCDs idle part of the dispatch)",
                   "kernels": rows}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print(f"{'kernel':44s} {'calls':>5s} {'ms/step':>8s} {'rd MiB/op':>9s} {'wr MiB/op':>9s} {'TB/s':>5s} {'HBM frac':>8s} {'VALU winstr':>11s} {'VALU frac':>9s} {'MHz':>5s}  bound")
    for r in rows:
        print(f"{r['kernel'][:44]:44s} {r['calls']:5d} {r['ms_per_step']:8.3f} {r['read_MiB_per_op']:9.3f} {r['write_MiB_per_op']:9.3f} {r['hbm_TBps']:5.2f} "
              f"{r['hbm_frac_of_8TBps']:8.3f} {r['valu_wave_instr']:11.4g} {r['valu_issue_frac']:9.3f} {r['sustained_mhz']:5d}  {r['bound']}")
    print(f"sum: {tot_ms:.2f} ms/step (single stream, under the counter pass), {tot_b:.1f} MiB/op HBM, {tot_i:.4g} VALU wave-instructions/step "
          f"= {tot_i / ops:.4g} per op")


if __name__ == "__main__":
    main()
