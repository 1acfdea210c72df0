#!/bin/bash
# A/B of two builds of the library on the same box: lib/libhebench_mi355x_backend.so against lib/alt_*.so (swapped in place).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
L=reference-seal-backend_amd/lib
run() { python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', d['value'], d['ms_per_step'])"; }
cp $L/libhebench_mi355x_backend.so /tmp/main.so
for rep in 1 2 3; do
  cp /tmp/main.so $L/libhebench_mi355x_backend.so && run main || exit 1
  cp $L/$1 $L/libhebench_mi355x_backend.so && run alt || exit 1
done
cp /tmp/main.so $L/libhebench_mi355x_backend.so
