#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5o; mkdir -p $O
for rep in 1 2; do
for ch in 1024 2048 4096 512; do
  timeout -k 10 300 python3 bench.py --config bfv_matmul --steps 5 --warmup 1 --cpu-sample 0 --parity-sample 1 --chunk $ch > $O/bfv_chunk_${ch}_$rep.json 2> $O/bfv_chunk_${ch}_$rep.err
  python3 -c "import json;j=json.load(open('$O/bfv_chunk_${ch}_$rep.json'));print('bfv_matmul chunk $ch rep $rep', j['ms_per_step'], j['parity']['checked_in_run'])"
done
done
