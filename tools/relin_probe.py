#!/usr/bin/env python3
"""Time he355_relinearize (+ rescale) of size-3 ciphertexts at the headline parameters (test tool).  usage: tools/relin_probe.py [n] [reps]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
be = importlib.import_module("reference-seal-backend_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, bits = 32768, [60] + [45] * 15 + [60]
g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=True, device=0)
L = g.L
g.set_relin_key_synthetic(7)
ct3 = g.alloc(n * 3 * L * N)
g.fill_uniform(ct3, n * 3 * L, list(range(L)), 11)
for rescale in (False, True):
    out = g.alloc(n * 2 * (L - 1 if rescale else L) * N)
    for it in range(reps + 2):
        if it == 2:
            g.sync(); t0 = time.perf_counter()
        if rescale:
            g.relinearize_rescale(L, n, ct3, out)
        else:
            g.relinearize(L, n, ct3, out)
    g.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"relinearize{'+rescale' if rescale else ''} n={n}: {dt * 1e3:.3f} ms per call, {n / dt:.0f} results/s")
