#!/bin/bash
# Collects the round's evidence on the GPU box: bench JSON, kernel-trace stats, PMC traffic passes.
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 3 --warmup 1 --profile-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- $B > gpurun_out/${TAG}_trace.log 2>&1
HE355_DUAL_STREAM=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -- python3 bench.py --steps 1 --warmup 0 --profile-mode > gpurun_out/${TAG}_fetch.log 2>&1
HE355_DUAL_STREAM=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -- python3 bench.py --steps 1 --warmup 0 --profile-mode > gpurun_out/${TAG}_write.log 2>&1
HE355_DUAL_STREAM=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq -- python3 bench.py --steps 1 --warmup 0 --profile-mode > gpurun_out/${TAG}_sq.log 2>&1
(cd tools && python3 hbm_traffic.py ../gpurun_out/${TAG}_fetch ../gpurun_out/${TAG}_write 1024 ../gpurun_out/${TAG}_hbm_traffic.json)
# the bench line is taken AFTER the traffic passes, so that its traffic_source quotes this round's measurement
cp gpurun_out/${TAG}_hbm_traffic.json profiles/${TAG}_hbm_traffic.json
python3 bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 tools/pmc_summary.py gpurun_out/${TAG}_fetch > gpurun_out/${TAG}_pmc_fetch_per_kernel.csv
python3 tools/pmc_summary.py gpurun_out/${TAG}_write > gpurun_out/${TAG}_pmc_write_per_kernel.csv
python3 tools/pmc_summary.py gpurun_out/${TAG}_sq > gpurun_out/${TAG}_pmc_sq_per_kernel.csv
cat gpurun_out/${TAG}_bench.json

# the other key-switch configurations: kernel-trace summaries (per-kernel ms per step) and their bench lines
for cfg in dot bfv_matmul mul_relin; do
  python3 bench.py --config $cfg --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_$cfg.json 2> gpurun_out/${TAG}_bench_$cfg.err
  HE355_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace_$cfg -- python3 bench.py --config $cfg --steps 2 --warmup 1 --profile-mode > gpurun_out/${TAG}_trace_$cfg.log 2>&1
  python3 tools/kstats.py gpurun_out/${TAG}_trace_$cfg 3 > gpurun_out/${TAG}_kernels_$cfg.txt
  cat gpurun_out/${TAG}_kernels_$cfg.txt
done
