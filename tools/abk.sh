#!/bin/bash
# Per-kernel single-stream times of library builds (usage on the GPU box: tools/abk.sh <tag> [<tag> ...]; "main" = the product).
# Timing experiments only: a variant's results may be wrong by design.  Selected with HE355_LIB_PATH, nothing is swapped.
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/reference-seal-backend_amd/lib
for tag in "$@"; do
  lib=$L/alt_$tag.so
  [ "$tag" = main ] && lib=$L/libhebench_mi355x_backend.so
  [ -f "$lib" ] || { echo "$tag: $lib missing"; exit 1; }
  rm -rf /tmp/ksx_$tag
  HE355_LIB_PATH=$lib HE355_DUAL_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksx_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --profile-mode > /tmp/ksx_$tag.log 2>&1 || { echo "$tag failed"; tail -5 /tmp/ksx_$tag.log; exit 1; }
  echo "== $tag"
  python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/ksx_$tag 4
done
