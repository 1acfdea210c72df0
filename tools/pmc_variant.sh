#!/bin/bash
# SQ / LDS counters of the headline step for library variants (usage on the GPU box: tools/pmc_variant.sh <tag> ...; "main" = product).
# PMC passes use --kernel-trace only (no other trace domains), single stream, one step.
cd /tmp && export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/reference-seal-backend_amd/lib
for tag in "$@"; do
  lib=$L/alt_$tag.so
  [ "$tag" = main ] && lib=$L/libhebench_mi355x_backend.so
  [ -f "$lib" ] || { echo "$tag: $lib missing"; exit 1; }
  rm -rf /tmp/pmcv_$tag
  HE355_LIB_PATH=$lib HE355_DUAL_STREAM=0 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmcv_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --profile-mode > /tmp/pmcv_$tag.log 2>&1 || { echo "$tag failed"; tail -5 /tmp/pmcv_$tag.log; exit 1; }
  echo "== $tag"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmcv_$tag | grep -E "^kernel|k_k3|k_k2|k_k1<0;ArF64|floor_cols"
done
