#!/bin/bash
# rotate_sum level sums on the second stream: parity (GPU suite), then same-box A/B against the old scattered gather (lib/alt_lsum1.so)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5z; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
A="HE355_LIB_PATH=$PWD/reference-seal-backend_amd/lib/alt_lsum1.so"
B="HE355_LIB_PATH=$PWD/reference-seal-backend_amd/lib/libhebench_mi355x_backend.so"
tools/ab_cfg.sh bfv_matmul "$A" "$B" | tee $O/ab_bfv.txt

