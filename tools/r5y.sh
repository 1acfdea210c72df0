#!/bin/bash
# mid-batch region: finer curve in the throughput shape and per-kernel tables at batch 8 / 16 / 32 / 64
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5y; mkdir -p $O
HE355_LATENCY_MAX=0 python3 tools/batch_curve.py --config mul_relin_rescale --batches 8 10 12 14 16 20 24 28 32 40 48 64 2>&1 | grep "^#" | tee $O/curve.txt
for b in 8 16 32 64; do
  echo "== batch $b" | tee -a $O/k.txt
  HE355_LATENCY_MAX=0 tools/ktrace.sh $O/k_$b.txt 4 bench.py --config mul_relin_rescale --batch $b --steps 3 --warmup 1 --profile-mode --cpu-sample 0 >> $O/k.txt 2>&1
done
cat $O/k.txt
