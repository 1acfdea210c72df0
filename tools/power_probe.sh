#!/bin/bash
# Samples clocks and power while the headline bench runs (usage: tools/power_probe.sh)
cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 40 --warmup 2 --profile-mode > gpurun_out/pp_bench.json 2> gpurun_out/pp_bench.err &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|fclk" | tr -s ' ' | head -8
  echo ---
  sleep 0.4
done > gpurun_out/pp_smi.txt 2>&1
wait $BP
cat gpurun_out/pp_smi.txt | head -40
grep -o '"ms_per_step": [0-9.]*' gpurun_out/pp_bench.json
