#!/bin/bash
# Kernels of ONE operate() call of a bridge descriptor (tools/bench_bridge.py --only <name>): two rocprofv3 kernel traces of the same
# pipeline with R1 and R2 timed operate() calls; the per-kernel difference / (R2 - R1) is what one call launches (encode, encrypt, key
# generation, decrypt, decode cancel out).  Usage on the GPU box: tools/bridge_ktrace.sh <out.txt> <sizes: default|bench> "<descriptor substring>"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; SIZES=$2; NAME=$3
R1=4; R2=24
cd /tmp && export TMPDIR=/tmp
for r in $R1 $R2; do
  D=/tmp/bk_$$_$r
  # Offline descriptors run max(3, reps/4) calls: scale so that both categories differ by 20 calls
  ( export HE355_DUAL_STREAM=0; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 "$R/tools/bench_bridge.py" --sizes $SIZES --only "$NAME" --reps $r --exact-reps --no-direct > $D.log 2>&1 ) || { tail -5 $D.log; exit 1; }
done
cd "$R" && python3 tools/kdiff.py /tmp/bk_$$_$R1 /tmp/bk_$$_$R2 $((R2 - R1)) "$NAME ($SIZES)" | tee "$OUT"
