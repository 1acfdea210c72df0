#!/usr/bin/env python3
"""Where a CKKS decode spends its time: he355_ckks_decode_slots (device kernels, HIP events) and the download, for n plaintexts at a given ring,
level and slot count.  Usage (GPU box): python tools/decode_probe.py [N] [depth] [n] [count]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
be = importlib.import_module("reference-seal-backend_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = be.Context(be.SCHEME_CKKS, N, bit_sizes=be.chain_bits(depth, 45), device=0)
L = g.L
for n, count in ((1, 1), (1, N // 2), (256, 1), (256, N // 2), (64, 1)):
    plain = g.alloc(n * L * N)
    g.fill_uniform(plain, n * L, list(range(L)), 9)
    out = g.alloc(n * count)
    for rep in range(3):
        g.sync()
        t0 = time.perf_counter()
        g.timer_begin()
        g.ckks_decode_slots(L, n, plain, 2.0 ** 45, [(0, count)], out)
        ms = g.timer_end()
        t1 = time.perf_counter()
        host = out.download()
        t2 = time.perf_counter()
        print(f"N={N} L={L} n={n} count={count} rep{rep}: kernels {ms:.3f} ms (call {1e3 * (t1 - t0):.3f} ms), download of {host.nbytes} B {1e3 * (t2 - t1):.3f} ms", flush=True)
    plain.free(); out.free()
g.close()
