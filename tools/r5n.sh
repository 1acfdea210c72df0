#!/bin/bash
# what the driver does at round end, in one call: smoke(), the default bench line; then a chunk sweep of the headline on the final tree
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5n; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 -c "import json;j=json.load(open('$O/bench_default.json'));r=j['roofline'];print('default', j['value'], j['ms_per_step'], r['bound'], r['frac'], r['valu']['frac'], r['valu']['sustained_mhz'], j['vs_baseline'], j['parity']['checked_in_run'])"
for ch in 1024 512 256; do
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --parity-sample 0 --chunk $ch > $O/chunk_${ch}_$rep.json 2> $O/chunk_${ch}_$rep.err
    python3 -c "import json;j=json.load(open('$O/chunk_${ch}_$rep.json'));print('chunk $ch rep $rep', j['ms_per_step'])"
  done
done
