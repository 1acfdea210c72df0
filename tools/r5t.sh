#!/bin/bash
# same-box A/B: k_k3 operand and own-digit rows with the default cache policy (product) vs non-temporal loads (lib/alt_rowsnt.so)
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5t; mkdir -p $O
L=$PWD/reference-seal-backend_amd/lib
for rep in 1 2 3; do
for arm in rows rowsnt; do
  if [ $arm = rowsnt ]; then export HE355_LIB_PATH=$L/alt_rowsnt.so; else unset HE355_LIB_PATH; fi
  for cfg in bfv_matmul dot mul_relin_rescale; do
    timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 --parity-sample 1 > $O/${cfg}_${arm}_$rep.json 2> $O/${cfg}_${arm}_$rep.err
    python3 -c "import json;j=json.load(open('$O/${cfg}_${arm}_$rep.json'));print('$cfg $arm $rep', j['ms_per_step'], j['parity']['checked_in_run'])"
  done
done
done
