#!/bin/bash
# Memory-latency / fifo counters of the headline step (single stream), per kernel.  usage on the GPU box: tools/pmc4.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
export HE355_DUAL_STREAM=0
B="python3 $R/bench.py --steps 1 --warmup 0 --profile-mode"
mkdir -p $R/gpurun_out
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/${TAG}_$i -- $B > /tmp/${TAG}_$i.log 2>&1 || { echo "set $i failed"; tail -3 /tmp/${TAG}_$i.log; continue; }
  python3 $R/tools/pmc_summary.py /tmp/${TAG}_$i > $R/gpurun_out/${TAG}_pmc4_$i.csv
  cut -c1-200 $R/gpurun_out/${TAG}_pmc4_$i.csv | head -7
done
