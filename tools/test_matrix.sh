#!/bin/bash
# Runs the GPU test-suite under every selectable code path (usage on the GPU box: tools/test_matrix.sh).
# Default settings first, then: the unfused floor steps (k_floor_rows finishes the mod-down), the host-side client, a single stream with
# a chunk size that divides nothing, the throughput shape for every batch size (HE355_LATENCY_MAX=0), the latency shape up to batch 64,
# the ct x ct tensor written by k_k1 instead of formed in k_k3 (HE355_C01_RECOMPUTE=0), chunks of 256 on the two-stream schedule,
# he355_rotate_sum / he355_rotate_each node by node instead of by grouped launches (HE355_LEVEL_WALK=0), the device pool off
# (HE355_POOL=0: hipMalloc / hipFree per he355_malloc / he355_free), every vector workload of the bridge spread over a two-device
# group (logical devices on this one GPU), the BEHZ multiply on SEAL's 61-bit auxiliary base (HE355_BEHZ_BASE=seal) and with its
# column passes in kernels of their own (HE355_BEHZ_FUSE=0), and every BFV product extending and transforming its own operands
# (HE355_BEHZ_HOIST=0: no operand transformed once for several results), and the latency shape with one launch per arithmetic engine
# and stage (HE355_DUAL_ENGINE=0) instead of both engines in one, and the unfused mod-down for every batch size (HE355_FUSE_MIN_BLOCKS=100000;
# HE355_K3_FUSE=0 above does the same through the older switch), and the fused mod-down plus one launch per engine for every throughput-shape
# batch, as before those two rules (HE355_FUSE_MIN_BLOCKS=0 HE355_DUAL_MAX_BLOCKS=0), and every prime -- the BEHZ auxiliary base included -- on
# the u64 engine (HE355_FORCE_U64=1).  (The kernel variants of rounds 2-3 -- HE355_XCHG, KSHARE, K2_SPLIT, K2_NEW=0, FC_NEW=0,
# FC_MERGE=0, K3_SHAPE, K3_STAGE, LAT_SIDE, SIDE_ALL -- were deleted in round 4: HISTORY.md.)
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
rc=0
for cfg in "HE355_NONE=1" "HE355_K3_FUSE=0" "HE355_DEVICE_CLIENT=0" "HE355_DUAL_STREAM=0 HE355_CHUNK=3" "HE355_LATENCY_MAX=0" "HE355_LATENCY_MAX=64" "HE355_C01_RECOMPUTE=0" "HE355_CHUNK=256" "HE355_LEVEL_WALK=0" "HE355_POOL=0" "HE355_NUM_DEVICES=2 HE355_LOGICAL_DEVICES=2" "HE355_BEHZ_BASE=seal" "HE355_BEHZ_FUSE=0" "HE355_BEHZ_BASE=seal HE355_BEHZ_FUSE=0" "HE355_BEHZ_HOIST=0" "HE355_DUAL_ENGINE=0" "HE355_FUSE_MIN_BLOCKS=100000" "HE355_FUSE_MIN_BLOCKS=0 HE355_DUAL_MAX_BLOCKS=0" "HE355_FORCE_U64=1" "HE355_K3_FOUR_WAVES_MAX=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 || rc=1
done
exit $rc
