#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
bash tools/bridge_ktrace.sh $O/k_logreg_offline.txt default "LogisticRegression_PolyD3 Offline"
