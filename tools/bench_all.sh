#!/bin/bash
# Every bench.py configuration on one MI355X (one JSON line each), then the multi-rank launch path rehearsed with two gloo
# ranks sharing the GPU (weak and strong scaling).  usage on the GPU box: tools/bench_all.sh <out-tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_bench_all.jsonl
: > $OUT
for cfg in mul_relin_rescale eltwise_mul dot bfv_matmul bfv_add; do
  echo "== $cfg"
  timeout -k 10 600 python3 bench.py --gpus 1 --config $cfg --steps 3 --warmup 1 2> gpurun_out/${TAG}_bench_$cfg.err | tee -a $OUT | cut -c1-400 || { tail -5 gpurun_out/${TAG}_bench_$cfg.err; exit 1; }
done
for sc in weak strong; do
  echo "== 2 gloo ranks on one GPU, $sc"
  HE355_BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 2 --warmup 1 --batch 256 --scaling $sc 2> gpurun_out/${TAG}_bench_2rank_$sc.err | tee -a $OUT | cut -c1-400 || { tail -5 gpurun_out/${TAG}_bench_2rank_$sc.err; exit 1; }
done
