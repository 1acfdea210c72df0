#!/usr/bin/env python3
"""One key switch (apply_galois, rotate_add) on 64 ciphertexts at BASELINE configs[4]'s parameters (N = 2^15, {60, 40, 40, 60}) through
the BFV pipeline and through the CKKS (NTT-domain, fused) pipeline: us per call.  Usage: python tools/ks_probe.py [bfv|ckks] [reps]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
be = importlib.import_module("reference-seal-backend_amd")
N, bits = 32768, [60, 40, 40, 60]
n = int(os.environ.get("KS_BATCH", "64"))
which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for scheme, name in ((be.SCHEME_BFV, "bfv"), (be.SCHEME_CKKS, "ckks")):
    if which not in ("both", name):
        continue
    g = be.Context(scheme, N, bit_sizes=bits, plain_bits=20 if scheme == be.SCHEME_BFV else 0, device=0)
    L = g.L
    a = g.alloc(n * 2 * L * N)
    out = g.alloc(n * 2 * L * N)
    g.fill_uniform(a, n * 2 * L, list(range(L)), 1)
    elt = g.galois_elt(1)
    g.set_galois_key_synthetic(elt, 5)
    g.set_latency_max(0)
    for _ in range(3):
        g.apply_galois(L, n, a, elt, out)
    g.sync()
    g.timer_begin()
    for _ in range(reps):
        g.apply_galois(L, n, a, elt, out)
    ms = g.timer_end()
    print(name, "apply_galois batch", n, "us per call", round(ms / reps * 1e3, 1))
    g.timer_begin()
    for _ in range(reps):
        g.rotate_add(L, n, a, 1, out, out)
    ms = g.timer_end()
    print(name, "rotate_add in place", n, "us per call", round(ms / reps * 1e3, 1))
    g.close()
