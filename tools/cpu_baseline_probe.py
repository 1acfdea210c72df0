#!/usr/bin/env python3
"""Thread scaling of the CPU stand-in (oracle/he_oracle.c) on THIS host, without a GPU: BASELINE configs[2]'s pipeline
(multiply -> relinearize -> rescale, N=2^15, L=16) on random operands, per thread count.  Same loop and scratch policy as
bench.py's cpu_baseline leg.  Test infrastructure only.
Usage: python tools/cpu_baseline_probe.py [threads ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as ho  # noqa: E402

N, bits = 32768, ho.chain_bits(16, 45)
o = ho.Context(ho.SCHEME_CKKS, N, bit_sizes=bits)
rng = np.random.default_rng(1)
cpus = ho.effective_cpus()
counts = [int(x) for x in sys.argv[1:]] or sorted({1, 2, 4, 8, 16, 32, 64, cpus["effective"]} & set(range(1, cpus["effective"] + 1)))
rk = o.random_kswitch_key(rng)
b = o.random_poly(rng, o.L, 2)[None]
one = None
print(json.dumps({"cpu_share": cpus, "N": N, "L": o.L}))
for t in counts:
    n = max(8 if t == 1 else 2 * t, int((one or 4.0) * t * 2.5))
    a = np.stack([o.random_poly(rng, o.L, 2) for _ in range(n)])
    o.batch_outer(ho.OP_MUL_RELIN_RESCALE, a[:t], b, rk, threads=t)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        o.batch_outer(ho.OP_MUL_RELIN_RESCALE, a, b, rk, threads=t)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    rate = n / best
    if t == 1:
        one = rate
    print(json.dumps({"threads": t, "results": n, "seconds": round(best, 3), "ops_per_sec": round(rate, 3),
                      "efficiency_vs_1_thread": round(rate / (one * t), 3) if one else None}), flush=True)
