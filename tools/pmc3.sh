#!/bin/bash
# SQ counter passes of the headline step (single stream), summarised per kernel.  usage on the GPU box: tools/pmc3.sh <tag> ["ENV=.. ENV=.."]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
[ -n "$2" ] && export $2
export HE355_DUAL_STREAM=0
B="python3 $R/bench.py --steps 1 --warmup 0 --profile-mode"
mkdir -p $R/gpurun_out
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/${TAG}_a -- $B > /tmp/${TAG}_a.log 2>&1 || { tail -5 /tmp/${TAG}_a.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/${TAG}_b -- $B > /tmp/${TAG}_b.log 2>&1 || { tail -5 /tmp/${TAG}_b.log; exit 1; }
python3 $R/tools/pmc_summary.py /tmp/${TAG}_a > $R/gpurun_out/${TAG}_pmc_a.csv
python3 $R/tools/pmc_summary.py /tmp/${TAG}_b > $R/gpurun_out/${TAG}_pmc_b.csv
cut -c1-220 $R/gpurun_out/${TAG}_pmc_a.csv | head -8
cut -c1-220 $R/gpurun_out/${TAG}_pmc_b.csv | head -8
