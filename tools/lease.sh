#!/bin/bash
# One parameterised GPU lease: runs the named steps on the GPU box (under gpurun), every step's output under gpurun_out/<tag>/.
#   tools/lease.sh <tag> <step> [<step> ...]        (run through: gpurun --timeout 1200 -- 'bash tools/lease.sh r6b tests rccl1 ...')
# Steps stop at the first failure (a GPU step that failed has told us something: no further GPU step in the same call).
# Steps:
#   tests[:<files / args>]  python -m pytest <files: default tests> -m gpu -x -q     -> pytest.log
#   shapes                  the bench-shape module only                             -> pytest_shapes.log
#   bench[:<args>]          python bench.py <args>                                  -> bench_<n>.json
#   rccl1                   bench.py --gpus 1 --force-dist (one-rank nccl = RCCL)   -> rccl_1rank.json
#   ranks5                  5 gloo ranks on the one card (the box admits 6 GPU processes and the launcher's agent is one): strong 1024,
#                           then both scalings in one invocation                    -> 5ranks_strong.json, 5ranks_both.json
#   dry8                    8 gloo ranks, no GPU: --dry-run, default scalings       -> 8ranks_dry.json
#   bridge[:<args>]         tools/bench_bridge.py <args> (default --sizes both)     -> bridge_phases.jsonl
#   ktrace:<substr>         rocprofv3 kernel trace (timestamps) of a bridge case    -> kt_<n>/
#   profile:<config>:<ops>  tools/profile_cfg.sh <tag> <config> <results per step>  -> gpurun_out/<tag>_<config>_*
#   matrix                  tools/test_matrix.sh                                    -> test_matrix.txt
#   sh:<command>            any command                                             -> sh_<n>.log
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R" || exit 1
TAG=$1; shift
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
k=0
run() { # run <logfile> <command...>: elapsed seconds appended, tail shown
  local log=$1; shift
  local t0=$SECONDS
  "$@" > "$log" 2> "$log.err"; local rc=$?
  echo "[lease] $(basename "$log"): rc=$rc in $((SECONDS - t0)) s" | tee -a "$OUT/steps.txt"
  tail -c 1200 "$log"; [ $rc -ne 0 ] && tail -c 1500 "$log.err"
  return $rc
}
for step in "$@"; do
  k=$((k + 1))
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $name in
    tests)   run "$OUT/pytest.log" python -m pytest ${arg:-tests} -m gpu -x -q || exit 1 ;;
    shapes)  run "$OUT/pytest_shapes.log" python -m pytest tests/test_gpu_bench_shapes.py -m gpu -x -q || exit 1 ;;
    bench)   run "$OUT/bench_$k.json" python bench.py $arg || exit 1 ;;
    rccl1)   run "$OUT/rccl_1rank.json" python bench.py --gpus 1 --force-dist --steps 5 || exit 1 ;;
    ranks5)  HE355_BENCH_BACKEND=gloo run "$OUT/5ranks_strong.json" python bench.py --gpus 5 --scaling strong --batch 1024 --steps 2 || exit 1
             HE355_BENCH_BACKEND=gloo run "$OUT/5ranks_both.json" python bench.py --gpus 5 --batch 96 --steps 2 || exit 1 ;;
    dry8)    HE355_BENCH_BACKEND=gloo run "$OUT/8ranks_dry.json" python bench.py --gpus 8 --dry-run || exit 1 ;;
    bridge)  run "$OUT/bridge_$k.log" python tools/bench_bridge.py ${arg:---sizes both} --out "$OUT/bridge_phases.jsonl" || exit 1 ;;
    ktrace)  ( cd /tmp && export TMPDIR=/tmp HE355_DUAL_STREAM=0 && run "$OUT/kt_$k.log" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt_$k" -- \
               python3 "$R/tools/bench_bridge.py" --sizes default --only "$arg" --reps 6 --exact-reps --no-direct ) || exit 1 ;;
    profile) run "$OUT/profile_${arg%%:*}.log" bash tools/profile_cfg.sh "$TAG" "${arg%%:*}" "${arg#*:}" || exit 1 ;;
    matrix)  run "$OUT/test_matrix.txt" bash tools/test_matrix.sh || exit 1 ;;
    sh)      run "$OUT/sh_$k.log" bash -c "$arg" || exit 1 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
