#!/bin/bash
# run-time tuning sweep of the headline step (usage on the GPU box: tools/sweep.sh): chunk size, K3 op-groups per block, stagger
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env "$@" timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --profile-mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$*', d['value'], d['ms_per_step'])"; }
for c in 96 128 192 256 384 512; do run HE355_CHUNK=$c; done
for og in 1 2 4 8; do run HE355_K3_OG=$og; done
run HE355_STAGGER=0
run HE355_DUAL_STREAM=0
run HE355_NONE=1
