#!/bin/bash
# Runs the GPU test-suite under every selectable code path (usage on the GPU box: tools/test_matrix.sh).
# Default settings first, then: unfused floor steps, the one-wave-per-SIMD K3 shape, register-prefetch K3, host-side client,
# single stream with a chunk size that divides nothing, the split k_k2, the unmerged floor corrections, round 2's k_k2 / k_floor_cols (HE355_K2_NEW=0, HE355_FC_NEW=0), k_k2n's XCD-local block order, the throughput shape for every batch size (HE355_LATENCY_MAX=0) and the latency shape up to batch 64 (also with the second engine's launches on a side stream, HE355_LAT_SIDE=1), the ct x ct tensor written by k_k1 instead of formed in k_k3 (HE355_C01_RECOMPUTE=0), round 3's earlier chunk size of 256 with the two-stream schedule, and every vector workload of the bridge spread over a
# two-device group (logical devices on this one GPU).  Compile-time variants (HE355_XCHG, HE355_KSHARE) are covered by
# tools/variant_matrix.sh.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
rc=0
for cfg in "HE355_NONE=1" "HE355_K3_FUSE=0" "HE355_K3_SHAPE=2414" "HE355_K3_STAGE=0" "HE355_DEVICE_CLIENT=0" "HE355_DUAL_STREAM=0 HE355_CHUNK=3" "HE355_K2_SPLIT=1" "HE355_FC_MERGE=0" "HE355_K2_NEW=0" "HE355_FC_NEW=0" "HE355_K2_XCD=1" "HE355_LATENCY_MAX=0" "HE355_LATENCY_MAX=64" "HE355_LATENCY_MAX=64 HE355_LAT_SIDE=1" "HE355_C01_RECOMPUTE=0" "HE355_SIDE_ALL=1" "HE355_CHUNK=256" "HE355_NUM_DEVICES=2 HE355_LOGICAL_DEVICES=2"; do
  echo "== $cfg"
  env $cfg timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 || rc=1
done
exit $rc
