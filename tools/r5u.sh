#!/bin/bash
# DIAGNOSTIC: k_k2n with and without its pattern-row stores -- kernel time and the clock the dispatch holds (GRBM_GUI_ACTIVE / 8 / time)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5u; mkdir -p $O
export HE355_DUAL_STREAM=0
for arm in product k2nostore; do
  if [ $arm = k2nostore ]; then export HE355_LIB_PATH=$PWD/reference-seal-backend_amd/lib/alt_k2nostore.so; else unset HE355_LIB_PATH; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_$arm -- python3 bench.py --profile-mode --steps 1 --warmup 0 > $O/sq_$arm.log 2>&1
  python3 tools/pmc_summary.py $O/sq_$arm > $O/sq_$arm.csv
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/sq_$arm.csv")):
    if "k_k2n" in r["kernel"] or "k_k3<ArF64" in r["kernel"]:
        us=float(r["total_us"]); cyc=float(r["GRBM_GUI_ACTIVE"])/8
        print("$arm", r["kernel"], "ms", round(us/1e3,2), "MHz", round(cyc/us), "VALU issue", round(float(r["SQ_INSTS_VALU"])*4/(1024*cyc),3), "wait_any/wave_cycles", round(float(r["SQ_WAIT_ANY"])/float(r["SQ_WAVE_CYCLES"]),3))
PY
done
