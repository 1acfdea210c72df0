"""Pins for the CPU oracle (no GPU).  The reference holds no fixtures, so the pins are:
prime-chain known answers (tests/golden/primes.json, generated with sympy), algebraic identities, and
Dec(Eval(Enc(x))) == f(x) with the oracle's own keys."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "primes.json")))


def test_get_primes_golden(oracle):
    L = oracle.lib()
    for e in GOLD["get_primes"] + GOLD["aux61"]:
        bits = e.get("bits", 61)
        want = [int(x, 16) for x in e["primes"]]
        buf = np.zeros(len(want), dtype=np.uint64)
        assert L.ho_get_primes(2 * e["N"], bits, len(want), oracle._p(buf)) == len(want)
        assert [int(x) for x in buf] == want


def test_chain_rule_golden(oracle):
    """{60, b x (depth-1), 60} -> primes, slot order as CoeffModulus::Create (seal_context.cpp:79-89)."""
    for e in GOLD["chains"]:
        bs = oracle.chain_bits(e["depth"], e["bits"])
        assert bs == e["bit_sizes"]
        ctx = oracle.Context(oracle.SCHEME_CKKS, e["N"], bit_sizes=bs)
        assert ctx.moduli == [int(x, 16) for x in e["primes"]]
        assert ctx.K == e["depth"] + 1 and ctx.L == e["depth"]


def test_batching_plain_modulus_golden(oracle):
    for e in GOLD["batching"]:
        if e["N"] < 8192:
            continue
        ctx = oracle.Context(oracle.SCHEME_BFV, e["N"], bit_sizes=[60, 40, 60], plain_bits=20)
        assert ctx.t == e["t"]


def test_security_gate(oracle):
    """BASELINE config 1 as worded (N=4096, {60,..,60}) is rejected under tc128."""
    with pytest.raises(ValueError):
        oracle.Context(oracle.SCHEME_BFV, 4096, bit_sizes=[60, 60], plain_bits=20)
    with pytest.raises(ValueError):
        oracle.Context(oracle.SCHEME_CKKS, 32768, bit_sizes=[60] + [50] * 16 + [60])


def test_minimal_root(oracle):
    ctx = oracle.Context(oracle.SCHEME_CKKS, 8192, bit_sizes=[60, 45, 60])
    for i, q in enumerate(ctx.moduli):
        r = ctx.root(i)
        assert pow(r, 8192, q) == q - 1
        # minimal among all primitive 2N-th roots: brute force over odd powers
        g2 = r * r % q
        cur, best = r, r
        for _ in range(8192):
            best = min(best, cur)
            cur = cur * g2 % q
        assert best == r
        rp = ctx.root_powers(i)
        assert int(rp[0]) == 1 and int(rp[1]) == pow(r, 4096, q)  # bitrev(1) = N/2


def _schoolbook_negacyclic(a, b, q):
    n = len(a)
    out = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            v = a[i] * b[j]
            if k >= n:
                out[k - n] = (out[k - n] - v) % q
            else:
                out[k] = (out[k] + v) % q
    return out


def test_ntt_is_negacyclic_convolution(oracle):
    N = 64
    buf = np.zeros(3, dtype=np.uint64)
    oracle.lib().ho_get_primes(2 * N, 60, 2, oracle._p(buf))
    oracle.lib().ho_get_primes(2 * N, 33, 1, oracle._p(buf[2:]))
    primes = [int(x) for x in buf]
    ctx = oracle.Context(oracle.SCHEME_CKKS, N, primes=primes)
    rng = np.random.default_rng(1)
    for i, q in enumerate(primes):
        a = rng.integers(0, q, N, dtype=np.uint64)
        b = rng.integers(0, q, N, dtype=np.uint64)
        fa, fb = ctx.ntt(i, a), ctx.ntt(i, b)
        prod = np.array([int(x) * int(y) % q for x, y in zip(fa, fb)], dtype=np.uint64)
        got = ctx.intt(i, prod)
        want = _schoolbook_negacyclic([int(x) for x in a], [int(x) for x in b], q)
        assert [int(x) for x in got] == want
        # NTT form = evaluations at psi^(2*bitrev(i)+1)
        r = ctx.root(i)
        k = 5
        brk = int(format(k, "06b")[::-1], 2)
        ev = sum(int(a[n]) * pow(r, (2 * brk + 1) * n, q) for n in range(N)) % q
        assert int(fa[k]) == ev


@pytest.mark.parametrize("N", [1024, 32768])
def test_ntt_roundtrip(oracle, N):
    ctx = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=[60, 45, 60], sec128=False)
    rng = np.random.default_rng(2)
    for i, q in enumerate(ctx.moduli):
        a = rng.integers(0, q, N, dtype=np.uint64)
        f = ctx.ntt(i, a)
        assert f.max() < q
        assert np.array_equal(ctx.intt(i, f), a)


def test_galois_forms_agree(oracle):
    N = 256
    ctx = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=[40, 30, 40], sec128=False)
    rng = np.random.default_rng(3)
    elts = ctx.galois_elts_all()
    assert len(elts) == 2 * (8 - 1) + 1 and elts[0] == 2 * N - 1 and elts[1] == 3
    assert ctx.galois_elt(1) == 3 and ctx.galois_elt(2) == 9 and ctx.galois_elt(0) == 2 * N - 1
    assert ctx.galois_elt(-1) == pow(3, N // 2 - 1, 2 * N)
    for elt in elts[:5]:
        a = rng.integers(0, ctx.moduli[0], N, dtype=np.uint64)
        lhs = ctx.ntt(0, ctx.apply_galois_poly(0, elt, False, a))
        rhs = ctx.apply_galois_poly(0, elt, True, ctx.ntt(0, a))
        assert np.array_equal(lhs, rhs)


# ---------------------------------------------------------------------------------------------------
# Dec(Eval(Enc)) with the oracle's own keys
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ckks_env(oracle):
    N, scale = 2048, 2.0 ** 40
    ctx = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=[60, 40, 40, 60], sec128=False)
    sk = ctx.keygen_secret(11)
    pk = ctx.keygen_public(sk, 12)
    rk = ctx.keygen_relin(sk, 13)
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, N // 2)
    y = rng.uniform(-1, 1, N // 2)
    cx = ctx.encrypt(pk, oracle.ckks_encode(ctx, x, scale), 21)
    cy = ctx.encrypt(pk, oracle.ckks_encode(ctx, y, scale), 22)
    return dict(ctx=ctx, sk=sk, pk=pk, rk=rk, x=x, y=y, cx=cx, cy=cy, scale=scale)


def _dec(oracle, env, ct, scale):
    ctx = env["ctx"]
    return oracle.ckks_decode(ctx, ctx.decrypt_phase(ct, env["sk"]), scale)


def test_ckks_encrypt_add(oracle, ckks_env):
    e = ckks_env
    assert np.allclose(_dec(oracle, e, e["cx"], e["scale"]).real, e["x"], atol=1e-6)
    s = e["ctx"].add(e["cx"], e["cy"])
    assert np.allclose(_dec(oracle, e, s, e["scale"]).real, e["x"] + e["y"], atol=1e-6)


def test_ckks_multiply_size3_relin_rescale(oracle, ckks_env):
    e = ckks_env
    ctx = e["ctx"]
    c3 = ctx.multiply_ntt(e["cx"], e["cy"])
    want = e["x"] * e["y"]
    # the reference decrypts un-relinearized size-3 results (ckks eltwise .cpp:342-344)
    assert np.allclose(_dec(oracle, e, c3, e["scale"] ** 2).real, want, atol=1e-5)
    c2 = ctx.relinearize(c3, e["rk"])
    assert np.allclose(_dec(oracle, e, c2, e["scale"] ** 2).real, want, atol=1e-5)
    c2r = ctx.rescale(c2)
    assert c2r.shape == (2, ctx.L - 1, ctx.N)
    new_scale = e["scale"] ** 2 / ctx.moduli[ctx.L - 1]
    assert np.allclose(_dec(oracle, e, c2r, new_scale).real, want, atol=1e-5)
    # batched pipeline == step-by-step
    a = e["cx"][None]
    b = e["cy"][None]
    got = ctx.batch_op(oracle.OP_MUL_RELIN_RESCALE, a, [0], b, [0], e["rk"])
    assert np.array_equal(got[0], c2r)


def test_ckks_rotate_and_accumulate(oracle, ckks_env):
    """rotate_vector semantics + the log-tree of accumulateCKKS (seal_context.cpp:331-339)."""
    e = ckks_env
    ctx = e["ctx"]
    for step in (1, 4, -2):
        elt = ctx.galois_elt(step)
        gk = ctx.keygen_galois(e["sk"], elt, 100 + step)
        r = ctx.apply_galois(e["cx"], elt, gk)
        assert np.allclose(_dec(oracle, e, r, e["scale"]).real, np.roll(e["x"], -step), atol=1e-5)
    n = 8
    acc = e["cx"]
    for i in range(3):
        elt = ctx.galois_elt(1 << i)
        gk = ctx.keygen_galois(e["sk"], elt, 200 + i)
        acc = ctx.add(acc, ctx.apply_galois(acc, elt, gk))
    got = _dec(oracle, e, acc, e["scale"]).real
    assert abs(got[0] - e["x"][:n].sum()) < 1e-4


def test_ckks_lower_level_keyswitch(oracle, ckks_env):
    """Key switching below the top level uses digits j < L and primes {0..L-1, special}."""
    e = ckks_env
    ctx = e["ctx"]
    c2r = ctx.rescale(ctx.relinearize(ctx.multiply_ntt(e["cx"], e["cy"]), e["rk"]))
    s1 = e["scale"] ** 2 / ctx.moduli[ctx.L - 1]
    c3 = ctx.multiply_ntt(c2r, c2r)
    c2 = ctx.relinearize(c3, e["rk"])
    want = (e["x"] * e["y"]) ** 2
    assert np.allclose(_dec(oracle, e, c2, s1 * s1).real, want, atol=1e-3)


@pytest.fixture(scope="module")
def bfv_env(oracle):
    N = 2048
    ctx = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=[50, 40, 50], plain_bits=20, sec128=False)
    codec = oracle.BatchCodec(N, ctx.t)
    sk = ctx.keygen_secret(31)
    pk = ctx.keygen_public(sk, 32)
    rk = ctx.keygen_relin(sk, 33)
    rng = np.random.default_rng(5)
    x = rng.integers(-500, 500, N)
    y = rng.integers(-500, 500, N)
    cx = ctx.encrypt(pk, codec.encode(x), 41)
    cy = ctx.encrypt(pk, codec.encode(y), 42)
    return dict(ctx=ctx, codec=codec, sk=sk, pk=pk, rk=rk, x=x, y=y, cx=cx, cy=cy)


def _bfv_dec(env, ct):
    ctx = env["ctx"]
    v = env["codec"].decode(ctx.bfv_decode_phase(ctx.decrypt_phase(ct, env["sk"])))
    t = ctx.t
    return np.where(v > t // 2, v - t, v)


def test_bfv_encrypt_add(bfv_env):
    e = bfv_env
    assert np.array_equal(_bfv_dec(e, e["cx"]), e["x"])
    assert np.array_equal(_bfv_dec(e, e["ctx"].add(e["cx"], e["cy"])), e["x"] + e["y"])


def test_bfv_multiply_behz(bfv_env):
    e = bfv_env
    ctx = e["ctx"]
    c3 = ctx.bfv_multiply(e["cx"], e["cy"])
    t = ctx.t
    want = (e["x"] * e["y"]) % t
    want = np.where(want > t // 2, want - t, want)
    assert np.array_equal(_bfv_dec(e, c3), want)
    c2 = ctx.relinearize(c3, e["rk"])
    assert np.array_equal(_bfv_dec(e, c2), want)


def test_bfv_rotate_rows_columns(bfv_env):
    e = bfv_env
    ctx = e["ctx"]
    N = ctx.N
    elt = ctx.galois_elt(3)
    r = ctx.apply_galois(e["cx"], elt, ctx.keygen_galois(e["sk"], elt, 51))
    x = e["x"].reshape(2, N // 2)
    assert np.array_equal(_bfv_dec(e, r).reshape(2, N // 2), np.roll(x, -3, axis=1))
    elt = ctx.galois_elt(0)
    r = ctx.apply_galois(e["cx"], elt, ctx.keygen_galois(e["sk"], elt, 52))
    assert np.array_equal(_bfv_dec(e, r).reshape(2, N // 2), x[::-1])


def _golden_pipeline():
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_pipeline_vectors", os.path.join(here, "golden", "make_pipeline_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, json.load(open(os.path.join(here, "golden", "pipeline_sha256.json")))


def test_pipeline_regression_fixture(oracle):
    """tests/golden/pipeline_sha256.json: the oracle reproduces the committed checksums of its own outputs on seeded inputs (a
    regression pin of this repository's arithmetic across rounds, not a SEAL vector); the GPU path is held to the same file in
    tests/test_gpu_parity.py::test_pipeline_regression_fixture_gpu."""
    gen, want = _golden_pipeline()
    for case in gen.CASES:
        assert gen.expected(oracle, case) == want[case["name"]], case["name"]
