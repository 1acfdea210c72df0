#!/usr/bin/env python3
"""Which core clock a batch-1 key-switch chain runs at: back-to-back calls, and single calls after idle gaps like those a harness leaves
between the operate() calls of a Latency run (the idle chip sleeps at ~150 MHz; how fast does it come back?).
Usage (GPU box): python tools/latency_clock_probe.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
be = importlib.import_module("reference-seal-backend_amd")

N, bits = 8192, [60, 40, 60]
g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
L = g.L
g.set_relin_key_synthetic(3)
g.set_galois_key_synthetic(g.galois_elt(1), 5)
d_a, acc = g.alloc(2 * L * N), g.alloc(2 * L * N)
g.fill_uniform(d_a, 2 * L, list(range(L)), 1)
g.fill_uniform(acc, 2 * L, list(range(L)), 4)


def chain(k):
    for _ in range(k):
        g.rotate_add(L, 1, d_a, 1, acc, acc)


for _ in range(20):
    chain(1)
g.sync()
# (a) a warm chain of 200 calls with the probe wave beside it
g.clock_probe_begin(8000)
g.timer_begin()
chain(200)
ms = g.timer_end()
mhz, cov = g.clock_probe_end()
print(f"warm chain of 200 batch-1 key switches: {1e3 * ms / 200:.1f} us per call, probe clock {mhz:.0f} MHz over {1e3 * cov:.2f} ms", flush=True)
# (b) eight key switches (one DotProduct-sized operate()) after an idle gap
for gap_ms in (0, 1, 5, 20, 100, 500):
    best = []
    for rep in range(5):
        g.sync()
        time.sleep(gap_ms / 1e3)
        t0 = time.perf_counter()
        g.timer_begin()
        chain(8)
        ms = g.timer_end()
        wall = (time.perf_counter() - t0) * 1e3
        best.append((ms, wall))
    best.sort()
    print(f"idle gap {gap_ms:4d} ms -> 8 key switches: GPU {best[len(best) // 2][0] * 1e3 / 8:.1f} us per call (median of 5; min {best[0][0] * 1e3 / 8:.1f}), host wall {best[len(best) // 2][1]:.3f} ms", flush=True)
g.close()
