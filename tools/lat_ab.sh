#!/bin/bash
# batch-1/2/8 latency of the headline op under several settings (usage on the GPU box: tools/lat_ab.sh "ENV=.." "ENV=.. ENV=.." ...), 2 rounds
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "$@"; do
    for n in 1 8; do
      ( export $cfg; timeout -k 10 120 python3 $R/tools/latency_probe.py $n 40 2>&1 | grep "sync each" | sed "s/^/$cfg batch $n: /" )
    done
  done
done
