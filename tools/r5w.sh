#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5w; mkdir -p $O
for rep in 1 2; do
for stg in 0 6 12 24 48 96; do
  for cfg in mul_relin_rescale bfv_matmul; do
    HE355_K3_STAGGER_TEST=$stg timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 --parity-sample 1 > $O/${cfg}_${stg}_$rep.json 2> $O/${cfg}_${stg}_$rep.err
    python3 -c "import json;j=json.load(open('$O/${cfg}_${stg}_$rep.json'));print('$cfg stagger $stg rep $rep', j['ms_per_step'], j['parity']['checked_in_run'])"
  done
done
done
