"""The oracle held to a second, independent derivation: tests/golden/exact_vectors.json was produced by the exact big-integer
model (tests/golden/exact_model.py: CRT composition, exact floors, Kronecker products; no RNS shortcuts, no oracle code, prime
chains from the sympy restatement).  The oracle must reproduce every expected output bit for bit on the same seeded inputs.
The GPU suite holds the HIP path to the same file (tests/test_gpu_parity.py::test_exact_model_fixture_gpu)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import exact_inputs as xi  # noqa: E402

FIX = json.load(open(os.path.join(HERE, "golden", "exact_vectors.json")))


def arr(x):
    return np.ascontiguousarray(np.array(x, dtype=np.uint64))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def case_inputs(name):
    """numpy views of the seeded inputs of a fixture case (the generator used the same functions)"""
    f = FIX[name]
    primes = [int(p, 16) for p in f["primes"]]
    N, seed, Ltop = f["N"], f["seed"], len(primes) - 1
    ge = f["galois_elts"]
    d = dict(primes=primes, N=N, Ltop=Ltop, a=arr(xi.ciphertext(seed, 1, primes, Ltop, 2, N)), b=arr(xi.ciphertext(seed, 2, primes, Ltop, 2, N)),
             rk=arr(xi.kswitch_key(seed, 3, primes, Ltop, N)),
             gk={int(g): arr(xi.kswitch_key(seed, 10 + n, primes, Ltop, N)) for n, g in enumerate((ge["1"], ge["-1"], ge["4"], ge["conj"]))})
    if f["scheme"] == "bfv":
        d["c3"] = arr(xi.ciphertext(seed, 4, primes, Ltop, 3, N))
    return f, d


def check(f, name, got):
    e = f["expected"][name]
    assert list(got.shape) == e["shape"], (name, got.shape)
    assert [[[int(v) for v in res[:3]] for res in poly] for poly in got] == e["head"], name  # first coefficients (diagnosis)
    assert sha(got) == e["sha256"], name


def run_ops(f, d, ops):
    """ops: object with the evaluator calls (oracle context here, the HIP path in the GPU suite); yields (name, result)"""
    ge = f["galois_elts"]
    a, b, rk, gk = d["a"], d["b"], d["rk"], d["gk"]
    yield "add", ops.add(a, b)
    if f["scheme"] == "ckks":
        c3 = ops.multiply(a, b)
        yield "multiply", c3
        rl = ops.relinearize(c3, rk)
        yield "multiply_relin", rl
        yield "multiply_relin_rescale", ops.rescale(rl)
        yield "rescale_size3", ops.rescale(c3)
        if d["Ltop"] >= 3:
            yield "relinearize_one_level_down", ops.relinearize(np.ascontiguousarray(c3[:, :d["Ltop"] - 1]), rk)
    else:
        yield "relinearize", ops.relinearize(d["c3"], rk)
        m3 = ops.multiply(a, b)  # BEHZ
        yield "bfv_multiply", m3
        yield "bfv_multiply_relin", ops.relinearize(m3, rk)
        yield "rotate_columns", ops.apply_galois(a, ge["conj"], gk[ge["conj"]])
    yield "rotate_1", ops.apply_galois(a, ge["1"], gk[ge["1"]])
    yield "rotate_3_naf", ops.apply_galois(ops.apply_galois(a, ge["-1"], gk[ge["-1"]]), ge["4"], gk[ge["4"]])


class OracleOps:
    def __init__(self, o):
        self.o = o

    def add(self, a, b): return self.o.add(a, b)
    def multiply(self, a, b): return self.o.multiply_ntt(a, b) if self.o.scheme == 2 else self.o.bfv_multiply(a, b)
    def relinearize(self, c3, rk): return self.o.relinearize(c3, rk)
    def rescale(self, ct): return self.o.rescale(ct)
    def apply_galois(self, ct, elt, key): return self.o.apply_galois(ct, elt, key)


@pytest.mark.parametrize("name", list(FIX))
def test_oracle_reproduces_the_exact_model(oracle, name):
    f, d = case_inputs(name)
    sid = oracle.SCHEME_CKKS if f["scheme"] == "ckks" else oracle.SCHEME_BFV
    o = oracle.Context(sid, f["N"], bit_sizes=f["bits"], plain_bits=20 if f["scheme"] == "bfv" else 0, sec128=False)
    assert [int(q) for q in o.moduli] == d["primes"]                       # CoeffModulus::Create, from the sympy restatement
    assert f["scheme"] == "ckks" or int(o.t) == f["plain_modulus"]
    assert [o.root(i) for i in range(len(d["primes"]))] == [int(p, 16) for p in f["psi"]]  # the minimal primitive 2N-th roots
    assert {s: o.galois_elt(int(s)) for s in ("1", "-1", "4")} == {s: f["galois_elts"][s] for s in ("1", "-1", "4")}
    seen = set()
    for opname, got in run_ops(f, d, OracleOps(o)):
        check(f, opname, got)
        seen.add(opname)
    assert seen == set(f["expected"])  # every pinned output was produced
    # the NAF rotation through the oracle's own rotate_internal restatement lands on the same ciphertext
    assert sha(o.rotate(d["a"], 3, {g: k for g, k in d["gk"].items() if g != o.galois_elt(3)})) == f["expected"]["rotate_3_naf"]["sha256"]


# ---- the sizes the bench runs (tests/golden/exact_vectors_big.json, make_exact_vectors_big.py) ------------------------------------
BIG_PATH = os.path.join(HERE, "golden", "exact_vectors_big.json")
BIG = json.load(open(BIG_PATH)) if os.path.exists(BIG_PATH) else {}


def big_case_inputs(name):
    """numpy inputs of a big fixture case (the generator used the same vectorised functions, checked there against the scalar ones)"""
    f = BIG[name]
    primes = [int(p, 16) for p in f["primes"]]
    N, seed, Ltop = f["N"], f["seed"], len(primes) - 1
    d = dict(primes=primes, N=N, Ltop=Ltop, a=xi.ciphertext_np(seed, 1, primes, Ltop, 2, N), b=xi.ciphertext_np(seed, 2, primes, Ltop, 2, N))
    if "multiply_relin" in f["expected"]:
        d["rk"] = xi.kswitch_key_np(seed, 3, primes, Ltop, N)
    if "rotate_1" in f["expected"] or f["scheme"] == "bfv":
        d["g1"] = f["galois_elts"]["1"]
        d["gk1"] = xi.kswitch_key_np(seed, 10, primes, Ltop, N)
    if f["scheme"] == "bfv":
        d["rk"] = xi.kswitch_key_np(seed, 3, primes, Ltop, N)
        d["gconj"] = f["galois_elts"]["conj"]
        d["gkc"] = xi.kswitch_key_np(seed, 13, primes, Ltop, N)
    return f, d


def test_numpy_input_twins_agree_with_the_scalar_functions():
    q = (1 << 60) - (1 << 18) + 1
    for seed, tag, mod, n in ((0xE0C1, 1003, q, 300), (7, 3, 35184371138561, 100)):
        assert xi.uniform_poly_np(seed, tag, mod, n).tolist() == xi.uniform_poly(seed, tag, mod, n)
    primes = [q, 35184371138561, 1152921504606584833]
    assert xi.ciphertext_np(5, 2, primes, 2, 2, 16).tolist() == xi.ciphertext(5, 2, primes, 2, 2, 16)
    assert xi.kswitch_key_np(5, 3, primes, 2, 16).tolist() == xi.kswitch_key(5, 3, primes, 2, 16)


@pytest.mark.parametrize("name", [c["name"] for c in xi.BIG_CASES if c["scheme"] == "bfv"])
def test_oracle_reproduces_the_exact_model_at_bench_sizes_bfv(oracle, name):
    """BASELINE configs[4]'s parameters (N = 2^15, {60, 40, 40, 60}, t = 786433): BEHZ multiply -> relinearize -> rotate_rows(1), the
    chain of bfv row .cpp:515-531, and rotate_columns -- the oracle against the exact model's integer statement, bit for bit."""
    if name not in BIG:
        pytest.skip("fixture case not generated (tests/golden/make_exact_vectors_big.py)")
    f, d = big_case_inputs(name)
    o = oracle.Context(oracle.SCHEME_BFV, f["N"], bit_sizes=f["bits"], plain_bits=20)
    assert [int(q) for q in o.moduli] == d["primes"] and int(o.t) == f["plain_modulus"]
    assert o.galois_elt(1) == d["g1"]
    m3 = o.bfv_multiply(d["a"], d["b"])
    check(f, "bfv_multiply", m3)
    rl = o.relinearize(m3, d["rk"])
    check(f, "bfv_multiply_relin", rl)
    check(f, "bfv_multiply_relin_rotate_rows_1", o.apply_galois(rl, d["g1"], d["gk1"]))
    check(f, "rotate_columns", o.apply_galois(d["a"], d["gconj"], d["gkc"]))


@pytest.mark.parametrize("name", [c["name"] for c in xi.BIG_CASES if c["scheme"] == "ckks"])
def test_oracle_reproduces_the_exact_model_at_bench_sizes(oracle, name):
    """N = 2^15 with the headline chain {60, 45 x 15, 60} (multiply, multiply -> relinearize, -> rescale, one rotation) and N = 2^14,
    {60, 45 x 7, 60} (multiply): the oracle on the exact model's ciphertexts, bit for bit."""
    if name not in BIG:
        pytest.skip("fixture case not generated (tests/golden/make_exact_vectors_big.py)")
    f, d = big_case_inputs(name)
    o = oracle.Context(oracle.SCHEME_CKKS, f["N"], bit_sizes=f["bits"])
    assert [int(q) for q in o.moduli] == d["primes"]
    assert [o.root(i) for i in range(len(d["primes"]))] == [int(p, 16) for p in f["psi"]]
    exp = f["expected"]
    c3 = o.multiply_ntt(d["a"], d["b"])
    check(f, "multiply", c3)
    if "multiply_relin" in exp:
        rl = o.relinearize(c3, d["rk"])
        check(f, "multiply_relin", rl)
        check(f, "multiply_relin_rescale", o.rescale(rl))
    if "rotate_1" in exp:
        assert o.galois_elt(1) == d["g1"]
        check(f, "rotate_1", o.apply_galois(d["a"], d["g1"], d["gk1"]))
