#!/usr/bin/env python3
"""Ring-in-LDS key switch (csrc/he355_kernels_lds.hip) against the HBM shapes, at the reference's default rings:
one dependent chain of key switches (rotate_add in place: what accumulateCKKS issues, /root/reference/src/engine/seal_context.cpp:331-339),
multiply -> relinearize and multiply -> relinearize -> rescale, for batch 1 .. 256.  us per call from HIP events over a chain of calls
on the context's stream (each call depends on the one before it: latency, not overlap).

Usage (GPU box): python tools/lds_probe.py [--n 8192] [--bits 60,40,60] [--batches 1,2,4,...] [--reps 50]"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
be = importlib.import_module("reference-seal-backend_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--bits", default="60,40,60")
    ap.add_argument("--batches", default="1,2,4,8,16,32,64,128,256")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    bits = [int(x) for x in a.bits.split(",")]
    g = be.Context(be.SCHEME_CKKS, a.n, bit_sizes=bits, sec128=False, device=0)
    L, N = g.L, a.n
    g.set_relin_key_synthetic(3)
    g.set_galois_key_synthetic(g.galois_elt(1), 5)
    print(f"# N = {N}, bits {bits} (L = {L}), us per call, chain of {a.reps} dependent calls; shapes: lds = ring-in-LDS (2 launches), hbm = the library's "
          f"rule without it (latency shape up to 2^17 / N ciphertexts, then unfused / fused throughput shapes)")
    print(f"# {'batch':>5s} {'rotate_add lds':>15s} {'hbm':>9s} {'mul_relin lds':>15s} {'hbm':>9s} {'mul_relin_rescale lds':>22s} {'hbm':>9s}")
    for n in [int(x) for x in a.batches.split(",")]:
        d_a, d_b = g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
        acc = g.alloc(n * 2 * L * N)
        out2 = g.alloc(n * 2 * max(1, L - 1) * N)
        g.fill_uniform(d_a, n * 2 * L, list(range(L)), 1)
        g.fill_uniform(d_b, n * 2 * L, list(range(L)), 2)
        g.fill_uniform(acc, n * 2 * L, list(range(L)), 4)
        pw = be.Context.pairwise()
        row = []
        for op in ("rot", "mr", "mrr"):
            for lds in (1 << 20, 0):
                g.set_lds_max(lds)

                def call():
                    if op == "rot":
                        g.rotate_add(L, n, d_a, 1, acc, acc)
                    elif op == "mr":
                        g.multiply_relin(L, n, d_a, d_b, pw, acc)
                    else:
                        g.multiply_relin(L, n, d_a, d_b, pw, out2, rescale=True)
                if op == "mrr" and L < 2:
                    row.append(float("nan"))
                    continue
                for _ in range(3):
                    call()
                g.sync()
                g.timer_begin()
                for _ in range(a.reps):
                    call()
                ms = g.timer_end()
                row.append(ms / a.reps * 1e3)
        print(f"  {n:5d} {row[0]:15.1f} {row[1]:9.1f} {row[2]:15.1f} {row[3]:9.1f} {row[4]:22.1f} {row[5]:9.1f}", flush=True)
        for b in (d_a, d_b, acc, out2):
            b.free()
    g.close()


if __name__ == "__main__":
    main()
