#!/bin/bash
# A/B of library builds on the same box (usage on the GPU box: tools/ab.sh <tag> [<tag> ...]; "main" = the product).
# Variants are built beside the product (make -C reference-seal-backend_amd/csrc VARIANT=<tag> DEFS="-D...") and selected
# with HE355_LIB_PATH: the product file is never touched.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
L=$PWD/reference-seal-backend_amd/lib
run() {
  local tag=$1 lib=$L/alt_$1.so
  [ "$tag" = main ] && lib=$L/libhebench_mi355x_backend.so
  [ -f "$lib" ] || { echo "$tag: $lib missing"; return 1; }
  HE355_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --profile-mode 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', d['value'], d['ms_per_step'])"
}
for rep in 1 2 3; do
  for tag in "$@"; do run "$tag" || exit 1; done
done
