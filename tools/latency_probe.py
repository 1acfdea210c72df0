#!/usr/bin/env python3
"""Batch-1 run of the headline operation for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/latency_probe.py)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
be = importlib.import_module("reference-seal-backend_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = be.Context(be.SCHEME_CKKS, 32768, bit_sizes=be.chain_bits(16, 45), device=0)
L, N = g.L, g.N
g.set_relin_key_synthetic(7)
a, b, o = g.alloc(n * 2 * L * N), g.alloc(2 * L * N), g.alloc(n * 2 * (L - 1) * N)
g.fill_uniform(a, n * 2 * L, list(range(L)), 1)
g.fill_uniform(b, 2 * L, list(range(L)), 2)
ix = be.Context.outer(0, n, 0, 1)
for _ in range(3):
    g.multiply_relin(L, n, a, b, ix, o, rescale=True)
g.sync()
t0 = time.perf_counter()
for _ in range(reps):
    g.multiply_relin(L, n, a, b, ix, o, rescale=True)
    g.sync()
print("ms per call, sync each:", (time.perf_counter() - t0) / reps * 1e3)
t0 = time.perf_counter()
for _ in range(reps):
    g.multiply_relin(L, n, a, b, ix, o, rescale=True)
g.sync()
print("ms per call, one sync:", (time.perf_counter() - t0) / reps * 1e3)
g.close()
