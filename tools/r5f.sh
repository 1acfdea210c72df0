#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for cfg in mul_relin_rescale dot bfv_matmul; do
  timeout -k 10 600 python3 bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 > $O/bench_${cfg}.json 2> $O/bench_${cfg}.err
  python3 -c "import json;j=json.load(open('$O/bench_${cfg}.json'));print('$cfg', j['value'], j['ms_per_step'], j['parity']['checked_in_run'])"
done
HE355_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bfv -- python3 bench.py --config bfv_matmul --steps 2 --warmup 1 --profile-mode > $O/trace_bfv.log 2>&1
python3 tools/kstats.py $O/trace_bfv 3 > $O/kernels_bfv_matmul.txt; cat $O/kernels_bfv_matmul.txt
python3 tools/bench_bridge.py --sizes default --reps 20 --only "LogisticRegression" --out $O/bridge_logreg.jsonl > $O/bridge_logreg.log 2>&1
python3 -c "
import json
for l in open('$O/bridge_logreg.jsonl'):
    j=json.loads(l); print(j['descriptor'], j['operate_ms'])"
bash tools/bridge_ktrace.sh $O/k_logreg_offline.txt default "LogisticRegression_PolyD3 Offline"
