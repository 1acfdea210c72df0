#!/bin/bash
# per-kernel single-stream times of an alternative library build (timing experiments; results may be wrong by design)
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/reference-seal-backend_amd/lib
cp $L/libhebench_mi355x_backend.so /tmp/main.so && cp $L/$1 $L/libhebench_mi355x_backend.so
HE355_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksx -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
cp /tmp/main.so $L/libhebench_mi355x_backend.so
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/ksx 4
