#!/bin/bash
# round-5 GPU job A: u64 fold micro-benchmark, the new bench line, counters for configs[3] / configs[4] / headline
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/u64_fold tools/micro/u64_fold.hip
timeout -k 10 300 $O/u64_fold 30000000 > $O/u64_fold.txt 2>&1
cat $O/u64_fold.txt
timeout -k 10 600 python3 bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
python3 -c "import json;j=json.load(open('$O/bench.json'));print(j['value'], j['ms_per_step'], j['roofline']['bound'], j['roofline']['valu'], j['roofline']['clock_probe'], j['vs_baseline'])"
timeout -k 10 300 python3 bench.py --config eltwise_mul --batch 16 --b1 16 --steps 50 --warmup 5 > $O/bench_eltwise_16x16.json 2> $O/bench_eltwise_16x16.err
timeout -k 10 300 python3 bench.py --config eltwise_mul --steps 50 --warmup 5 > $O/bench_eltwise_256x1.json 2> $O/bench_eltwise_256x1.err
tools/profile_cfg.sh r05 dot 64
tools/profile_cfg.sh r05 bfv_matmul 64
tools/profile_cfg.sh r05 mul_relin_rescale 1024
