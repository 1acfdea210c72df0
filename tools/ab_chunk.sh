#!/bin/bash
# alternating whole-step A/B of run-time settings (usage on the GPU box: tools/ab_chunk.sh "ENV=.." "ENV=.." ...), 5 rounds
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3 4 5; do
  for cfg in "$@"; do
    ( export $cfg; timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --profile-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', d['value'], d['ms_per_step'])" )
  done
done
