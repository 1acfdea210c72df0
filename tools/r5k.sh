#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p $O
timeout -k 10 700 python3 tools/fuzz_parity.py ${SECS:-400} ${SEED:-5002} > $O/fuzz_${SEED:-5002}.txt 2>&1 || { tail -5 $O/fuzz_${SEED:-5002}.txt; exit 1; }
tail -3 $O/fuzz_${SEED:-5002}.txt
grep -c " ok$" $O/fuzz_${SEED:-5002}.txt; grep -c MISMATCH $O/fuzz_${SEED:-5002}.txt || true
