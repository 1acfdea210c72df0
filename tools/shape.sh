#!/bin/bash
# K3 block-shape sweep (usage: tools/shape.sh 2414 1818 ...): bench value with the default dual-stream schedule, then
# per-kernel durations from a single-stream kernel trace (dual-stream overlap stretches individual kernels).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sh in "$@"; do
  HE355_K3_SHAPE=$sh timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --profile-mode > gpurun_out/shape_$sh.json 2> gpurun_out/shape_$sh.err || { echo "shape $sh failed"; tail -3 gpurun_out/shape_$sh.err; exit 1; }
  echo "shape=$sh $(grep -o '"value": [0-9.]*' gpurun_out/shape_$sh.json) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/shape_$sh.json)"
  HE355_DUAL_STREAM=0 HE355_K3_SHAPE=$sh timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shape_$sh -- python3 bench.py --steps 2 --warmup 1 --profile-mode > gpurun_out/shape_$sh.log 2>&1 || { echo "trace $sh failed"; exit 1; }
  python3 tools/kstats.py gpurun_out/shape_$sh 3
done
