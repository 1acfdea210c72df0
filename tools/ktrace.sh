#!/bin/bash
# Per-kernel table (rocprofv3 --kernel-trace --stats) of a command, single stream.  Usage on the GPU box:
#   tools/ktrace.sh <out.txt> <steps incl. warm-up> <python script and arguments ...>
# e.g. tools/ktrace.sh gpurun_out/k.txt 4 bench.py --config bfv_matmul --steps 3 --warmup 1 --profile-mode
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; STEPS=$2; shift 2
D=/tmp/ktrace_$$
cd /tmp && export TMPDIR=/tmp
( export HE355_DUAL_STREAM=0; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 "$R/$1" "${@:2}" > $D.log 2>&1 ) || { tail -5 $D.log; exit 1; }
cd "$R" && python3 tools/kstats.py $D $STEPS | tee "$OUT"
