#!/bin/bash
# alternating A/B of run-time settings on another bench configuration (usage on the GPU box: tools/ab_cfg.sh <config> "ENV=.." "ENV=.." ...), 3 rounds
cd "$GRAFT_REPO_ROOT" || exit 1
CFG=$1; shift
for rep in 1 2 3; do
  for cfg in "$@"; do
    ( export $cfg; timeout -k 10 300 python3 bench.py --config $CFG --steps 4 --warmup 1 --profile-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$CFG $cfg', d['value'], d['ms_per_step'])" )
  done
done
