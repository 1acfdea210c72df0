#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p $O
timeout -k 10 700 python3 tools/fuzz_parity.py 420 5001 > $O/fuzz_5001.txt 2>&1 || { tail -5 $O/fuzz_5001.txt; exit 1; }
tail -3 $O/fuzz_5001.txt
grep -c " ok$" $O/fuzz_5001.txt; grep -c MISMATCH $O/fuzz_5001.txt || true
