#!/bin/bash
# Runs the whole GPU test-suite under every selectable code path (usage on the GPU box: tools/test_matrix.sh).  The compact version of
# this matrix that the driver's `pytest -m gpu` runs is tests/test_gpu_code_paths.py.
# Default settings first, then: the unfused floor steps everywhere (HE355_K3_FUSE=0: k_floor_rows finishes the mod-down), the fused
# mod-down for every throughput-shape batch with one launch per engine and eight-wave u64-engine blocks (HE355_K3_FUSE=all: no small-grid
# rule), the host-side client, a single stream with a chunk size that divides nothing, the throughput shape for every batch size
# (HE355_LATENCY_MAX=0), the latency shape up to batch 64, chunks of 256 on the two-stream schedule, he355_rotate_sum / he355_rotate_each
# node by node instead of by grouped launches (HE355_LEVEL_WALK=0), the device pool off (HE355_POOL=0), every vector workload of the bridge
# spread over a two-device group (logical devices on this one GPU), the BEHZ multiply on SEAL's 61-bit auxiliary base
# (HE355_BEHZ_BASE=seal), with its column passes in kernels of their own (HE355_BEHZ_FUSE=2), with every product extending and transforming
# its own operands (HE355_BEHZ_FUSE=1), with neither (0), the latency shape with one launch per arithmetic engine and stage
# (HE355_DUAL_ENGINE=0), every prime -- the BEHZ auxiliary base included -- on the u64 engine (HE355_FORCE_U64=1: the Shoup build of the device code), and the default engine
# assignment on the Shoup build (HE355_FORCE_U64=shoup; without it the 60-bit primes take the fold build).
# (Retired in round 5 with their recorded losers, HISTORY.md: HE355_STAGGER, K2_TSPLIT, K3_OG, C01_RECOMPUTE, LAT_SPLIT(_U64); the three
# thresholds DUAL_MAX_BLOCKS / FUSE_MIN_BLOCKS / K3_FOUR_WAVES_MAX became constants behind HE355_K3_FUSE=all; BEHZ_HOIST is bit 1 of BEHZ_FUSE.)
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
rc=0
# Round 6: rings up to N = 8192 take the ring-in-LDS shape for small batches (HE355_LDS_MAX); the settings that select HBM shapes switch it off.
# usage: tools/test_matrix.sh [first [count]] -- a slice of the settings (a gpurun call is capped at 20 minutes; the whole matrix takes about 30)
CFGS=("HE355_NONE=1" "HE355_LDS_MAX=0" "HE355_LDS_MAX=64" "HE355_K3_FUSE=0 HE355_LDS_MAX=0" "HE355_K3_FUSE=all HE355_LDS_MAX=0" "HE355_DEVICE_CLIENT=0" "HE355_DUAL_STREAM=0 HE355_CHUNK=3" "HE355_LATENCY_MAX=0 HE355_LDS_MAX=0" "HE355_LATENCY_MAX=64 HE355_LDS_MAX=0" "HE355_CHUNK=256" "HE355_LEVEL_WALK=0" "HE355_POOL=0" "HE355_NUM_DEVICES=2 HE355_LOGICAL_DEVICES=2" "HE355_BEHZ_BASE=seal" "HE355_BEHZ_FUSE=2" "HE355_BEHZ_BASE=seal HE355_BEHZ_FUSE=0" "HE355_BEHZ_FUSE=1" "HE355_DUAL_ENGINE=0 HE355_LDS_MAX=0" "HE355_FORCE_U64=1" "HE355_FORCE_U64=shoup" "HE355_FORCE_U64=shoup HE355_LDS_MAX=0")
FIRST=${1:-0}; COUNT=${2:-${#CFGS[@]}}
for cfg in "${CFGS[@]:$FIRST:$COUNT}"; do
  echo "== $cfg"
  env $cfg timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^FAILED|^ERROR| passed| failed" | tail -3 || rc=1
done
exit $rc
