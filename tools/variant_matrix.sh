#!/bin/bash
# GPU suite under the compile-time kernel variants (built beside the product, selected with HE355_LIB_PATH).
# Build first (CPU, here):  for v in "xl3 -DHE355_XCHG=3" "ks -DHE355_XCHG=3 -DHE355_KSHARE=1" "xl1 -DHE355_XCHG=1" "xl2 -DHE355_XCHG=2"; do set -- $v; make -C reference-seal-backend_amd/csrc VARIANT=$1 DEFS="${*:2}"; done
# round 3's compile-time switches the same way: "madasm -DHE355_MAD_ASM=1" "noaccrun -DHE355_ACC_RUN=0" "keyu64 -DHE355_KEY_EARLY_U64=1" "ke3 -DHE355_KEY_EARLY=3" "nolazy -DHE355_LAZY_U64=0"
# then on the GPU box: tools/variant_matrix.sh xl3 ks xl1 xl2 madasm noaccrun keyu64 ke3 nolazy
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
rc=0
for tag in "$@"; do
  lib=$PWD/reference-seal-backend_amd/lib/alt_$tag.so
  [ -f "$lib" ] || { echo "$tag: $lib missing"; rc=1; continue; }
  echo "== $tag"
  HE355_LIB_PATH=$lib timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 || rc=1
done
exit $rc
