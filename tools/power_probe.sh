#!/bin/bash
# Samples clocks and power (rocm-smi, ordinary user) every ~0.3 s while the headline bench runs a long timed region, and prints the busy
# samples (GPU use > 50 %): what the chip draws and which clocks it holds under this load.  usage (GPU box): tools/power_probe.sh [steps]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
STEPS=${1:-300}
mkdir -p gpurun_out
python3 bench.py --steps "$STEPS" --warmup 2 --profile-mode > gpurun_out/pp_bench.json 2> gpurun_out/pp_bench.err &
BP=$!
: > gpurun_out/pp_smi.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr -s ' \t' ' ' | tr '\n' '|' >> gpurun_out/pp_smi.txt
  echo >> gpurun_out/pp_smi.txt
  sleep 0.3
done
wait $BP
echo "# samples: $(wc -l < gpurun_out/pp_smi.txt); busy samples (GPU use >= 50 %):"
grep -E "GPU use \(%\): ([5-9][0-9]|100)" gpurun_out/pp_smi.txt | sed -e 's/GPU\[0\] : //g' | head -60
grep -o '"ms_per_step": [0-9.]*' gpurun_out/pp_bench.json | head -1
