#!/usr/bin/env python3
"""Generates tests/golden/exact_vectors_big.json: the exact big-integer model (exact_model.py) at the sizes the bench runs.

  * N = 2^15, key chain {60, 45 x 15, 60} (BASELINE configs[2], the headline): multiply, multiply -> relinearize,
    multiply -> relinearize -> rescale, and one rotation (Galois element of step 1) -- the kernel instances the headline uses
    (k_k2n<5>, k_k3<.., fused>, k_floor_colsn<5, merged>) are then held to the model with no oracle involved;
  * N = 2^14, key chain {60, 45 x 7, 60} (configs[1]): multiply.
  * BFV, N = 2^15, key chain {60, 40, 40, 60}, t = 786433 (configs[4]): BEHZ multiply, -> relinearize, -> rotate_rows(1), and
    rotate_columns of an input -- the kernel instances of the MatMultRow product (k_behz_*, the BFV key-switch tail, k_bfv_galois).

Same model, same seeded input functions as make_exact_vectors.py; the inputs are produced by their numpy twins
(exact_inputs.*_np, checked here against the pure-Python functions on samples) because a key is 18 M residues.
Pure Python cost: one key switch at N = 2^15 is 544 key-polynomial interpolations and 544 Kronecker products of 590 kB integers --
about 25 minutes for the whole file on one core.  Usage: python tests/golden/make_exact_vectors_big.py"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import exact_inputs as xi  # noqa: E402
from exact_model import Model  # noqa: E402
from make_exact_vectors import digest, galois_elt, head  # noqa: E402
from make_primes import coeff_modulus_create  # noqa: E402


def log(*a):
    print("[%s]" % time.strftime("%H:%M:%S"), *a, file=sys.stderr, flush=True)


def to_lists(a):
    return a.tolist()


def run_case(case):
    N, bits, seed = case["N"], case["bits"], case["seed"]
    primes = coeff_modulus_create(N, bits)
    K, Ltop = len(primes), len(primes) - 1
    bfv = case["scheme"] == "bfv"
    M = Model(N, primes, ntt_form=not bfv)
    # the recursive transform against its definition at a few evaluation points per prime (the full check is O(N^2))
    for i in range(K):
        c = xi.uniform_poly(seed, 999 + i, primes[i], N)
        v = M.R.ntt(i, c)
        pts = M.R.points(i)
        for k in (0, 1, N // 2 + 3, N - 1):
            acc = 0
            for cf in reversed(c):
                acc = (acc * pts[k] + cf) % primes[i]
            assert acc == v[k], "recursive NTT differs from its definition"
        assert M.R.intt(i, v) == c
    log(case["name"], "transforms checked")
    # the numpy twins of the input functions produce the documented values
    assert xi.uniform_poly_np(seed, 1000, primes[0], N)[:64].tolist() == xi.uniform_poly(seed, 1000, primes[0], 64 if False else N)[:64]
    a = to_lists(xi.ciphertext_np(seed, 1, primes, Ltop, 2, N))
    b = to_lists(xi.ciphertext_np(seed, 2, primes, Ltop, 2, N))
    out = {"N": N, "bits": bits, "scheme": case["scheme"], "seed": seed, "primes": [hex(p) for p in primes], "psi": [hex(p) for p in M.R.psi],
           "galois_elts": {"1": galois_elt(1, N), "conj": 2 * N - 1}, "expected": {}}
    exp = out["expected"]

    def put(name, ct):
        exp[name] = {"sha256": digest(ct), "shape": [len(ct), len(ct[0]), N], "head": head(ct)}
        log(case["name"], name, "done")

    if bfv:
        # the chain of bfv row .cpp:515-531 on the exact model: BEHZ multiply (integer statement, no auxiliary base), relinearize,
        # rotate_rows by one step; plus rotate_columns of an input (accumulateBFV's column swap, seal_context.cpp:308)
        t = case["plain_modulus"]
        out["plain_modulus"] = t
        m3 = M.bfv_multiply(a, b, t)
        put("bfv_multiply", m3)
        rk = to_lists(xi.kswitch_key_np(seed, 3, primes, Ltop, N))
        rl = M.relinearize(m3, rk)
        del rk
        put("bfv_multiply_relin", rl)
        gk = to_lists(xi.kswitch_key_np(seed, 10, primes, Ltop, N))
        put("bfv_multiply_relin_rotate_rows_1", M.apply_galois(rl, galois_elt(1, N), gk))
        del gk
        gc = to_lists(xi.kswitch_key_np(seed, 13, primes, Ltop, N))
        put("rotate_columns", M.apply_galois(a, 2 * N - 1, gc))
        return out
    c3 = M.multiply_ckks(a, b)
    put("multiply", c3)
    if "multiply_relin" in case["ops"]:
        rk = to_lists(xi.kswitch_key_np(seed, 3, primes, Ltop, N))
        rl = M.relinearize(c3, rk)
        del rk
        put("multiply_relin", rl)
        put("multiply_relin_rescale", M.rescale(rl))
    if "rotate_1" in case["ops"]:
        g1 = galois_elt(1, N)
        gk = to_lists(xi.kswitch_key_np(seed, 10, primes, Ltop, N))
        put("rotate_1", M.apply_galois(a, g1, gk))
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "exact_vectors_big.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    only = sys.argv[1:]
    for case in xi.BIG_CASES:
        if only and case["name"] not in only:
            continue
        t0 = time.time()
        doc[case["name"]] = run_case(case)
        log(case["name"], "%.1f s" % (time.time() - t0))
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)
