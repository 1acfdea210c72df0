#!/bin/bash
# Per-configuration counters on the GPU box (round 5): kernel-trace stats + FETCH_SIZE / WRITE_SIZE / SQ passes
# (each PMC pass in its own run, with --kernel-trace only) of ONE bench step of <config>, then the per-kernel table with the
# binding resource (tools/kernel_bounds.py).
# usage: tools/profile_cfg.sh <tag> <config> <results per step>      outputs: gpurun_out/<tag>_<config>_*
set -e
TAG=$1; CFG=$2; OPS=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG}_${CFG}
B="python3 bench.py --config $CFG --profile-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_trace -- $B --steps 2 --warmup 1 > ${O}_trace.log 2>&1
python3 tools/kstats.py ${O}_trace 3 > ${O}_kernels.txt
export HE355_DUAL_STREAM=0
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_fetch -- $B --steps 1 --warmup 0 > ${O}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_write -- $B --steps 1 --warmup 0 > ${O}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${O}_sq -- $B --steps 1 --warmup 0 > ${O}_sq.log 2>&1
(cd tools && python3 hbm_traffic.py ../${O}_fetch ../${O}_write $OPS ../${O}_hbm_traffic.json)
python3 tools/pmc_summary.py ${O}_sq > ${O}_pmc_sq_per_kernel.csv
python3 tools/kernel_bounds.py ${O}_fetch ${O}_write ${O}_sq $OPS --json ${O}_kernel_bounds.json > ${O}_kernel_bounds.txt
cat ${O}_kernel_bounds.txt
