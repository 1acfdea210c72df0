#!/bin/bash
# round-5 GPU job E: (1) 5-byte-row timing probe (WRONG results by construction, timing only) on configs[4]; (2) dual-launch threshold
# sweep on the reference-default descriptors; (3) LogReg offline after the load()-time constants
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
L=$PWD/reference-seal-backend_amd/lib
for arm in base row5 base row5; do
  if [ $arm = row5 ]; then export HE355_LIB_PATH=$L/alt_row5.so; else unset HE355_LIB_PATH; fi
  timeout -k 10 300 python3 bench.py --config bfv_matmul --steps 5 --warmup 1 --cpu-sample 0 --parity-sample 0 > $O/row5_$arm.json 2> $O/row5_$arm.err
  python3 -c "import json;j=json.load(open('$O/row5_$arm.json'));print('bfv_matmul $arm', j['ms_per_step'])"
done
for arm in base row5; do
  if [ $arm = row5 ]; then export HE355_LIB_PATH=$L/alt_row5.so; else unset HE355_LIB_PATH; fi
  HE355_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$arm -- python3 bench.py --config bfv_matmul --steps 2 --warmup 1 --profile-mode > $O/trace_$arm.log 2>&1
  python3 tools/kstats.py $O/trace_$arm 3 > $O/kernels_bfv_$arm.txt
done
paste $O/kernels_bfv_base.txt $O/kernels_bfv_row5.txt | cut -c1-220
unset HE355_LIB_PATH
python3 tools/bench_bridge.py --sizes default --reps 20 --out $O/bridge_default.jsonl > $O/bridge_default.log 2>&1
python3 - <<PY
import json
for l in open("$O/bridge_default.jsonl"):
    j=json.loads(l); print("default", j.get("name") or j.get("descriptor"), j.get("operate_ms") or j.get("operate_ms_median"))
PY
export HE355_LIB_PATH=$L/alt_dmenv.so
for dm in 1024 2048 4096 8192; do
  HE355_DUAL_MAX_BLOCKS=$dm python3 tools/bench_bridge.py --sizes default --reps 20 --out $O/bridge_dm$dm.jsonl > $O/bridge_dm$dm.log 2>&1
done
unset HE355_LIB_PATH
python3 - <<PY
import json
rows={}
for dm in (1024,2048,4096,8192):
    for l in open("$O/bridge_dm%d.jsonl"%dm):
        j=json.loads(l); k=j.get("name") or j.get("descriptor"); rows.setdefault(k,{})[dm]=j.get("operate_ms") or j.get("operate_ms_median")
for k,v in rows.items(): print(k, v)
PY
