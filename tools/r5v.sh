#!/bin/bash
# same-box A/B: rotations with addend -- k_k3 gathers c0 and adds addend0 (product) vs k_k1 does both and writes the row (lib/alt_noaddg.so)
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5v; mkdir -p $O
L=$PWD/reference-seal-backend_amd/lib
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bfv.py tests/test_api_bridge_gpu.py tests/test_gpu_code_paths.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2 3; do
for arm in addg noaddg; do
  if [ $arm = noaddg ]; then export HE355_LIB_PATH=$L/alt_noaddg.so; else unset HE355_LIB_PATH; fi
  for cfg in bfv_matmul dot mul_relin_rescale; do
    timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 --parity-sample 1 > $O/${cfg}_${arm}_$rep.json 2> $O/${cfg}_${arm}_$rep.err
    python3 -c "import json;j=json.load(open('$O/${cfg}_${arm}_$rep.json'));print('$cfg $arm $rep', j['ms_per_step'], j['parity']['checked_in_run'])"
  done
done
done
