#!/bin/bash
# K2 variants on one box: GPU suite, per-kernel single-stream times, whole-step A/B (old k_k2 / split with 4 waves / split with 3 waves per SIMD)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== split=0 (old k_k2)"; HE355_K2_SPLIT=0 tools/abk.sh main 2>&1 | grep -v "^== main"
echo "== split, 4 waves"; tools/abk.sh main 2>&1 | grep -v "^== main"
echo "== split, 3 waves"; tools/abk.sh k2w3 2>&1 | grep -v "^== k2w3"
for r in 1 2; do
HE355_K2_SPLIT=0 tools/ab.sh main | head -1 | sed 's/main/old/'
tools/ab.sh main | head -1
tools/ab.sh k2w3 | head -1
done
