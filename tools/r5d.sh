#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -25 $O/pytest.log
