#!/bin/bash
# alternating A/B of two builds of the library on one bench configuration, 3 rounds (usage on the GPU box:
#   tools/ab_lib.sh <config> <alt .so under reference-seal-backend_amd/lib> ["extra bench args"]); A = the product, B = the alternative build
cd "$GRAFT_REPO_ROOT" || exit 1
CFG=$1; ALT="$PWD/reference-seal-backend_amd/lib/$2"; EXTRA=$3
for rep in 1 2 3; do
  for which in product alt; do
    ( [ $which = alt ] && export HE355_LIB_PATH=$ALT; timeout -k 10 300 python3 bench.py --config $CFG --steps 4 --warmup 1 --profile-mode $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$CFG $which', d['value'], d['ms_per_step'])" )
  done
done
