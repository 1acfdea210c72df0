#!/bin/bash
# Latency-category shape of the headline op (batch 1 / 8 / 64, host-synchronised per call) + the batch-1 kernel trace.
# usage on the GPU box: tools/latency_round.sh <tag> ["ENV=.."]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
[ -n "$2" ] && export $2
OUT=$R/gpurun_out/${TAG}_latency.txt
: > $OUT
for n in 1 2 8 64; do
  echo "== batch $n ${2}" | tee -a $OUT
  timeout -k 10 120 python3 $R/tools/latency_probe.py $n 30 2>&1 | grep "ms per call" | tee -a $OUT
done
rm -rf /tmp/lat_$TAG
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lat_$TAG -- python3 $R/tools/latency_probe.py 1 20 > /tmp/lat_$TAG.log 2>&1
echo "== batch-1 kernels (avg us per launch)" | tee -a $OUT
python3 - <<PY | tee -a $OUT
import csv, glob, os
f = max(glob.glob('/tmp/lat_$TAG/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
tot = 0
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('he355::(anonymous namespace)::', '').split('(')[0]
    if 'fill_uniform' in n or 'key_' in n: continue
    calls = int(r['Calls']); avg = float(r['AverageNs']) / 1e3
    per_call = calls / 43.0  # 3 warm-up + 20 + 20 calls of the op
    tot += avg * per_call
    print(f"  {n[:48]:48s} launches/op {per_call:5.2f}  avg_us {avg:8.1f}")
print(f"  sum of kernel time per op: {tot:.1f} us")
PY
