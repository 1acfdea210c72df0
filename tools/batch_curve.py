#!/usr/bin/env python3
"""One-GPU batch curve: ops/s and us/op of a bench.py configuration at batch 16 ... 1024 (the strong-scaling proxy: under
`--scaling strong --batch 1024` each of 8 GPUs sees 128 ciphertext pairs, each of 4 sees 256, each of 2 sees 512;
reference batch axis: /root/reference/src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:325).  Same workload objects, keys and
fill as bench.py; per batch W warm-up steps then K timed steps (wall clock around the steps + stream sync, and HIP events).
Usage (GPU box): python tools/batch_curve.py [--config mul_relin_rescale dot] [--batches 16 32 ...] [--steps 5]"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", nargs="+", default=["mul_relin_rescale", "dot"])
ap.add_argument("--batches", nargs="+", type=int, default=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024])
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--N", type=int, default=0, help="ring degree in place of the configuration's (CKKS chains {60, 45 x (depth-1), 60})")
ap.add_argument("--depth", type=int, default=0, help="chain depth in place of the configuration's")
args = ap.parse_args()

be = importlib.import_module("reference-seal-backend_amd")
sharding = bench.load_sharding()
for cfg in args.config:
    W = bench.WORKLOADS[cfg]
    if args.N or args.depth:
        W = type(W.__name__, (W,), {"N": args.N or W.N, "depth": args.depth or W.depth, "bits": None})
    bits = W.bits or be.chain_bits(W.depth, W.coeff_bits)
    rows = []
    for b in args.batches:
        if cfg == "dot" and b > 256:
            steps = max(2, args.steps // 2)
        else:
            steps = args.steps
        ctx = be.Context(be.SCHEME_CKKS if W.scheme == "ckks" else be.SCHEME_BFV, W.N, bit_sizes=bits, plain_bits=W.plain_bits, device=0)
        wl = W(be, ctx, sharding.shard_outer_product(b, W.b1, 1, 0), args)
        wl.setup()
        for _ in range(args.warmup):
            wl.step()
        ctx.sync()
        ctx.timer_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        ev_ms = ctx.timer_end()
        ctx.sync()
        wall = time.perf_counter() - t0
        n = wl.n
        rec = {"config": cfg, "batch": b, "results_per_step": n, "steps": steps, "ms_per_step": round(wall / steps * 1e3, 4),
               "ms_per_step_hip_events": round(ev_ms / steps, 4), "us_per_op": round(wall / steps / n * 1e6, 2), "ops_per_sec": round(n * steps / wall, 1)}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        ctx.close()
    ref = rows[-1]["us_per_op"]
    for r in rows:
        print(f"# {cfg:18s} batch {r['batch']:5d}  {r['ms_per_step']:10.3f} ms/step  {r['us_per_op']:9.2f} us/op  {r['ops_per_sec']:10.1f} ops/s  "
              f"x{r['us_per_op'] / ref:5.2f} of the batch-{rows[-1]['batch']} per-op time", flush=True)
