#!/bin/bash
# Scaling curve on one node (the driver's job on an 8-GPU node; here for whoever has one): bench.py at 1, 2, 4, 8 GPUs, weak
# (1024 results per GPU) and strong (global batch 1024: north_star's ">= 6x at 8 GPUs") -> one JSONL, one line per run.
# usage: tools/scale.sh [out.jsonl] [gpu counts, default "1 2 4 8"]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale.jsonl}
COUNTS=${2:-"1 2 4 8"}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29600
for n in $COUNTS; do
  for mode in weak strong; do
    PORT=$((PORT + 1))
    if [ "$n" = 1 ]; then
      python3 bench.py --gpus 1 --steps 5 --warmup 1 --scaling $mode --batch 1024 --cpu-sample 0 >> "$OUT"
    else
      python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n --steps 5 --warmup 1 \
        --scaling $mode --batch 1024 --cpu-sample 0 >> "$OUT"
    fi
    tail -1 "$OUT" | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$n GPU(s) $mode:', j['value'], 'ops/s', j['ms_per_step'], 'ms/step', 'allreduce_of_ones', (j.get('collective') or {}).get('allreduce_of_ones'), 'per_rank_ms', j.get('per_rank_ms'))"
  done
done
python3 - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = {r["scaling"]: r["value"] for r in rows if r["n_gpus"] == 1}
for r in rows:
    print(f"{r['n_gpus']} GPU(s) {r['scaling']:6s}: {r['value']:10.1f} ops/s = {r['value'] / base[r['scaling']]:.2f}x of 1 GPU")
PY
