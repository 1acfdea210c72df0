#!/bin/bash
# One GPU call: the -m gpu suite on the product build, then per-kernel and whole-step A/B of the given variants.
# usage: tools/gpu_ab_round.sh <out-tag> <variant> [<variant> ...]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?
tail -3 gpurun_out/${TAG}_pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed (rc=$rc): stopping"; exit $rc; fi
tools/abk.sh main "$@" > gpurun_out/${TAG}_abk.log 2>&1 || { tail -5 gpurun_out/${TAG}_abk.log; exit 1; }
cat gpurun_out/${TAG}_abk.log
tools/ab.sh main "$@" > gpurun_out/${TAG}_ab.log 2>&1 || { tail -5 gpurun_out/${TAG}_ab.log; exit 1; }
cat gpurun_out/${TAG}_ab.log
exit $rc
