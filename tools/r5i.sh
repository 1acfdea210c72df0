#!/bin/bash
# round-5 GPU job I: the round's evidence with the final tree -- counters per configuration, the bench lines, the full test matrix
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
tools/profile_cfg.sh r05 mul_relin_rescale 1024 > $O/prof_headline.log 2>&1
tools/profile_cfg.sh r05 dot 64 > $O/prof_dot.log 2>&1
tools/profile_cfg.sh r05 bfv_matmul 64 > $O/prof_bfv.log 2>&1
for c in mul_relin_rescale dot bfv_matmul; do cp gpurun_out/r05_${c}_kernel_bounds.json profiles/; done   # the bench lines below quote this run's counters
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err
for cfg in mul_relin dot bfv_matmul eltwise_mul bfv_add; do
  timeout -k 10 600 python3 bench.py --config $cfg --steps 5 --warmup 1 > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
timeout -k 10 300 python3 bench.py --config eltwise_mul --batch 16 --b1 16 --steps 50 --warmup 5 > $O/bench_eltwise_mul_16x16.json 2> $O/bench_eltwise_mul_16x16.err
for f in headline mul_relin dot bfv_matmul eltwise_mul bfv_add eltwise_mul_16x16; do python3 -c "
import json;j=json.load(open('$O/bench_$f.json'));r=j['roofline'];print('$f', j['value'], j['ms_per_step'], r['bound'], r['frac'], (r['valu'] or {}).get('frac'), j['vs_baseline'], j['parity']['checked_in_run'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench3 -- python3 bench.py --steps 3 --warmup 1 --profile-mode > $O/trace_bench3.log 2>&1
cp $(ls $O/trace_bench3/*/*kernel_stats.csv | head -1) $O/kernel_stats_bench_steps3.csv
python3 tools/bench_bridge.py --sizes both --reps 20 --out $O/bridge_phases.jsonl > $O/bridge.log 2>&1
echo "== test matrix"
bash tools/test_matrix.sh > $O/test_matrix.txt 2>&1 || true
cat $O/test_matrix.txt
echo "== batch curve"
python3 tools/batch_curve.py > $O/batch_curve.txt 2>&1 || true
grep "^#" $O/batch_curve.txt
