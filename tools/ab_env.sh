#!/bin/bash
# A/B of run-time settings of ONE library on the same box: per-kernel single-stream times (rocprofv3 kernel trace), then whole-step
# times alternating between the settings.  usage on the GPU box: tools/ab_env.sh <out-tag> "<ENV=..>" "<ENV=.. ENV=..>" ...
# ("HE355_NONE=1" = the defaults).  Optional: HE355_LIB_PATH in the caller's environment selects a variant build.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${TAG}_abenv.log
: > $OUT
i=0
for cfg in "$@"; do
  i=$((i+1))
  rm -rf /tmp/abenv_$i
  # env(1) cannot sit between rocprofv3 and the program (the profiler's preload initialises the GPU): export in a subshell instead
  ( export $cfg HE355_DUAL_STREAM=0; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abenv_$i -- python3 $R/bench.py --steps 3 --warmup 1 --profile-mode > /tmp/abenv_$i.log 2>&1 ) || { echo "$cfg failed" | tee -a $OUT; tail -5 /tmp/abenv_$i.log | tee -a $OUT; exit 1; }
  echo "== $cfg (single stream, per kernel)" | tee -a $OUT
  python3 $R/tools/kstats.py /tmp/abenv_$i 4 | tee -a $OUT
done
cd $R
for rep in 1 2 3; do
  for cfg in "$@"; do
    ( export $cfg; timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --profile-mode 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', d['value'], d['ms_per_step'])" ) | tee -a $OUT || exit 1
  done
done
