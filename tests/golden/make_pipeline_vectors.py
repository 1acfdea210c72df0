#!/usr/bin/env python3
"""Regression fixtures for the hot path: SHA-256 of the oracle's outputs on seeded inputs, at small parameters.

These are NOT vectors of the reference (SEAL is not available offline, SURVEY.md section 8c: parity against SEAL stays unpinned);
they pin THIS repository's arithmetic across rounds: tests/test_oracle_kat.py recomputes them with the oracle on the CPU and
tests/test_gpu_parity.py with the HIP path on the MI355X, so neither can drift without the other noticing.
Usage: python tests/golden/make_pipeline_vectors.py > tests/golden/pipeline_sha256.json"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [
    dict(name="ckks_n1024", scheme="ckks", N=1024, bits=[50, 40, 40, 50], n=3, seed=20260101),
    dict(name="ckks_n4096_mixed_engines", scheme="ckks", N=4096, bits=[60, 45, 47, 30, 60], n=2, seed=20260102),
    dict(name="bfv_n2048", scheme="bfv", N=2048, bits=[54, 40, 54], plain_bits=20, n=2, seed=20260103),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def inputs(o, case):
    """Seeded operands and keys of a case (shared by the generator and the tests)."""
    rng = np.random.default_rng(case["seed"])
    L = len(case["bits"]) - 1
    a = np.stack([o.random_poly(rng, L, 2) for _ in range(case["n"])])
    b = np.stack([o.random_poly(rng, L, 2) for _ in range(case["n"])])
    rk = o.random_kswitch_key(rng)
    gk = o.random_kswitch_key(rng)
    return L, a, b, rk, gk


def expected(oracle, case):
    ckks = case["scheme"] == "ckks"
    o = oracle.Context(oracle.SCHEME_CKKS if ckks else oracle.SCHEME_BFV, case["N"], bit_sizes=case["bits"], plain_bits=case.get("plain_bits", 0), sec128=False)
    L, a, b, rk, gk = inputs(o, case)
    out = {"moduli": [int(q) for q in o.moduli]}
    elt = o.galois_elt(1)
    if ckks:
        relin = [o.relinearize(o.multiply_ntt(a[r], b[r]), rk) for r in range(case["n"])]
        out["multiply_relin"] = sha(np.stack(relin))
        out["multiply_relin_rescale"] = sha(np.stack([o.rescale(x) for x in relin]))
    else:
        out["bfv_multiply_relin"] = sha(np.stack([o.relinearize(o.bfv_multiply(a[r], b[r]), rk) for r in range(case["n"])]))
    out["add"] = sha(np.stack([o.add(a[r], b[r]) for r in range(case["n"])]))
    out["rotate_1"] = sha(np.stack([o.apply_galois(a[r], elt, gk) for r in range(case["n"])]))
    return out


if __name__ == "__main__":
    oracle = importlib.import_module("oracle")
    print(json.dumps({c["name"]: expected(oracle, c) for c in CASES}, indent=1))
