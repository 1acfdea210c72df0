#!/bin/bash
# same-box sweep of the two dual-launch thresholds (k_k3 / the others) on configs[3], configs[4] and the default descriptors
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
export HE355_LIB_PATH=$PWD/reference-seal-backend_amd/lib/alt_dmenv.so
for rep in 1 2; do
for arm in "1024 1024" "4096 1024" "4096 4096" "8192 1024"; do
  set -- $arm
  export HE355_DUAL_MAX_BLOCKS=$1 HE355_DUAL_MAX_OTHER=$2
  for cfg in dot bfv_matmul; do
    timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 --parity-sample 0 > $O/${cfg}_$1_$2_$rep.json 2> $O/${cfg}_$1_$2_$rep.err
    python3 -c "import json;j=json.load(open('$O/${cfg}_$1_$2_$rep.json'));print('$cfg k3<=$1 others<=$2 rep $rep:', j['ms_per_step'])"
  done
  if [ $rep = 1 ]; then
    python3 tools/bench_bridge.py --sizes default --reps 20 --no-direct --out $O/bridge_$1_$2.jsonl > $O/bridge_$1_$2.log 2>&1
  fi
done
done
python3 - <<PY
import json
rows={}
arms=[(1024,1024),(4096,1024),(4096,4096),(8192,1024)]
for a in arms:
    for l in open("$O/bridge_%d_%d.jsonl"%a):
        j=json.loads(l); rows.setdefault(j["descriptor"],{})[a]=j["operate_ms"]
print("%-50s"%"descriptor"+"".join("%12s"%("%d/%d"%a) for a in arms))
for k,v in rows.items(): print("%-50s"%k+"".join("%12.4f"%v[a] for a in arms))
PY
