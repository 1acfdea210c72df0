#!/usr/bin/env python3
"""Generates tests/golden/exact_vectors.json from the exact big-integer model (exact_model.py) — no oracle, no product code.

The fixture pins, bit for bit, what the oracle (CPU suite: tests/test_exact_model.py) and the HIP path (GPU suite:
tests/test_gpu_parity.py::test_exact_model_fixture_gpu) must produce for multiply, relinearize, rescale, the fused
multiply -> relinearize -> rescale sequence, lower-level key switching, Galois rotations (single element and the two-term
NAF rotation by 3), CKKS and BFV, and BFV's BEHZ ct x ct multiply (as the integer formula of exact_model.py).  Expected outputs are stored as SHA-256 of the little-endian u64 array plus the first
coefficients of every residue polynomial (for diagnosis).  Prime chains come from the sympy restatement of
CoeffModulus::Create (make_primes.py), not from the oracle.

It is NOT a vector of the reference (SEAL is not in this image): it is a second, independent derivation of the same
definitions, so an agreement pins the oracle's RNS shortcuts (floor directions, digit layout, evaluation order), not the
recollection of SEAL's definitions themselves.  Usage: python tests/golden/make_exact_vectors.py   (about two minutes)."""
import hashlib
import json
import os
import struct
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import exact_inputs as xi  # noqa: E402
from exact_model import Model  # noqa: E402
from make_primes import coeff_modulus_create, get_primes  # noqa: E402


def digest(nested):
    flat = []

    def walk(x):
        if isinstance(x, list) and x and isinstance(x[0], list):
            for y in x:
                walk(y)
        else:
            flat.extend(x)
    walk(nested)
    return hashlib.sha256(struct.pack("<%dQ" % len(flat), *flat)).hexdigest()


def head(ct):
    return [[[int(v) for v in res[:3]] for res in poly] for poly in ct]


def galois_elt(step, N):
    m = 2 * N
    if step > 0:
        return pow(3, step, m)
    return pow(3, N // 2 - (-step), m)


def run_case(case):
    N, bits = case["N"], case["bits"]
    primes = coeff_modulus_create(N, bits)
    K, Ltop = len(primes), len(primes) - 1
    ckks = case["scheme"] == "ckks"
    M = Model(N, primes, ntt_form=ckks)
    # the recursive transform against the definition, once per prime
    for i in range(K):
        c = xi.uniform_poly(case["seed"], 999 + i, primes[i], N)
        assert M.R.ntt(i, c) == M.R.ntt_by_definition(i, c), "recursive NTT differs from its definition"
        assert M.R.intt(i, M.R.ntt(i, c)) == c
    seed = case["seed"]
    a = xi.ciphertext(seed, 1, primes, Ltop, 2, N)
    b = xi.ciphertext(seed, 2, primes, Ltop, 2, N)
    rk = xi.kswitch_key(seed, 3, primes, Ltop, N)
    g1, gm1, g4 = galois_elt(1, N), galois_elt(-1, N), galois_elt(4, N)
    gk = {g: xi.kswitch_key(seed, 10 + n, primes, Ltop, N) for n, g in enumerate((g1, gm1, g4, 2 * N - 1))}
    out = {"N": N, "bits": bits, "scheme": case["scheme"], "seed": seed, "primes": [hex(p) for p in primes],
           "psi": [hex(p) for p in M.R.psi], "galois_elts": {"1": g1, "-1": gm1, "4": g4, "conj": 2 * N - 1}, "expected": {}}
    exp = out["expected"]

    def put(name, ct):
        exp[name] = {"sha256": digest(ct), "shape": [len(ct), len(ct[0]), N], "head": head(ct)}

    put("add", M.add(a, b))
    if ckks:
        c3 = M.multiply_ckks(a, b)
        put("multiply", c3)
        rl = M.relinearize(c3, rk)
        put("multiply_relin", rl)
        put("multiply_relin_rescale", M.rescale(rl))
        put("rescale_size3", M.rescale(c3))
        if Ltop >= 3:  # key switch below the top level: the key's first L digits, L data primes + the special prime
            low = [[r for r in p[:Ltop - 1]] for p in c3]
            put("relinearize_one_level_down", M.relinearize(low, rk))
    else:
        c3 = xi.ciphertext(seed, 4, primes, Ltop, 3, N)  # BFV: key switching works on any size-3 ciphertext (coefficient form)
        put("relinearize", M.relinearize(c3, rk))
        t = get_primes(2 * N, 20, 1)[0]  # PlainModulus::Batching(N, 20) (seal_context.cpp:118)
        out["plain_modulus"] = t
        m3 = M.bfv_multiply(a, b, t)     # BEHZ as integer arithmetic
        put("bfv_multiply", m3)
        put("bfv_multiply_relin", M.relinearize(m3, rk))
        put("rotate_columns", M.apply_galois(a, 2 * N - 1, gk[2 * N - 1]))
    r1 = M.apply_galois(a, g1, gk[g1])
    put("rotate_1", r1)
    # Evaluator::rotate_internal without a key for 3: NAF 3 = -1 + 4, applied in that order
    put("rotate_3_naf", M.apply_galois(M.apply_galois(a, gm1, gk[gm1]), g4, gk[g4]))
    return out


if __name__ == "__main__":
    doc = {}
    for case in xi.CASES:
        t0 = time.time()
        doc[case["name"]] = run_case(case)
        print(case["name"], "%.1f s" % (time.time() - t0), file=sys.stderr)
    with open(os.path.join(HERE, "exact_vectors.json"), "w") as f:
        json.dump(doc, f, indent=1)
