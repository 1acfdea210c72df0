"""GPU parity for the BFV rows of SURVEY.md §8(a): BEHZ ct x ct multiply, relinearize, rotate_rows / rotate_columns
(coefficient-form Galois + key switch), accumulateBFV — bit-exact against the oracle on identical inputs and keys."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (N, key-level bit sizes, plain bits)
    "n1024": (1024, [50, 40, 50], 20),
    "n4096_d3": (4096, [60, 40, 40, 60], 20),
    "n8192_default": (8192, [60, 40, 60], 20),      # the reference's BFV defaults (bfv eltwise .h:23-26)
    "n32768_d3": (32768, [60, 40, 40, 60], 20),     # BASELINE configs[4] parameters (bfv row .h:29-32)
}


@pytest.fixture(scope="module")
def be():
    mod = importlib.import_module("reference-seal-backend_amd")
    if mod.device_count() < 1:
        pytest.fail("no HIP device")
    return mod


@pytest.fixture(scope="module", params=list(CONFIGS))
def pair(request, be, oracle):
    N, bits, pb = CONFIGS[request.param]
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False)
    assert g.moduli == o.moduli and g.t == o.t
    yield g, o, np.random.default_rng(N)
    g.close()


def rand_cts(o, rng, n, L, size=2):
    return np.stack([o.random_poly(rng, L, size) for _ in range(n)])


def test_bfv_multiply_behz(pair, be):
    g, o, rng = pair
    L, N = g.L, g.N
    a, b = rand_cts(o, rng, 3, L), rand_cts(o, rng, 2, L)
    out = g.alloc(6 * 3 * L * N)
    g.bfv_multiply(L, 6, g.to_device(a), g.to_device(b), be.Context.outer(0, 3, 0, 2), out)
    got = out.download((6, 3, L, N))
    for i in range(3):
        for x in range(2):
            assert np.array_equal(got[i * 2 + x], o.bfv_multiply(a[i], b[x])), (i, x)


def test_bfv_multiply_chunks_cut_across_outer_product_rows(pair, be):
    """A chunk size that does not divide the operand-1 batch (chunks then start in the middle of an outer-product row)."""
    g, o, rng = pair
    L, N = g.L, g.N
    a, b = rand_cts(o, rng, 3, L), rand_cts(o, rng, 2, L)
    out = g.alloc(6 * 3 * L * N)
    g.set_chunk(3)
    try:
        g.bfv_multiply(L, 6, g.to_device(a), g.to_device(b), be.Context.outer(0, 3, 0, 2), out)
    finally:
        g.set_chunk(256)
    got = out.download((6, 3, L, N))
    for i in range(3):
        for x in range(2):
            assert np.array_equal(got[i * 2 + x], o.bfv_multiply(a[i], b[x])), (i, x)


def test_bfv_relinearize(pair, be):
    g, o, rng = pair
    L, N = g.L, g.N
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    ct3 = rand_cts(o, rng, 3, L, size=3)
    out = g.alloc(3 * 2 * L * N)
    g.relinearize(L, 3, g.to_device(ct3), out)
    got = out.download((3, 2, L, N))
    for r in range(3):
        assert np.array_equal(got[r], o.relinearize(ct3[r], rk)), r


def test_bfv_rotations_and_accumulate(pair, be):
    g, o, rng = pair
    L, N = g.L, g.N
    keys = {}
    for s in (1, 2, 4):
        e = o.galois_elt(s)
        keys[e] = o.random_kswitch_key(rng)
        g.set_galois_key(e, keys[e])
    col = 2 * N - 1
    keys[col] = o.random_kswitch_key(rng)
    g.set_galois_key(col, keys[col])
    a = rand_cts(o, rng, 2, L)
    da = g.to_device(a)
    out = g.alloc(2 * 2 * L * N)
    g.rotate(L, 2, da, 2, out)  # rotate_rows by 2
    got = out.download((2, 2, L, N))
    for r in range(2):
        assert np.array_equal(got[r], o.apply_galois(a[r], o.galois_elt(2), keys[o.galois_elt(2)]))
    g.apply_galois(L, 2, da, col, out)  # rotate_columns
    got = out.download((2, 2, L, N))
    for r in range(2):
        assert np.array_equal(got[r], o.apply_galois(a[r], col, keys[col]))
    # rotate_add: out = addend + rotate_rows / rotate_columns (in), also in place
    b = rand_cts(o, rng, 2, L)
    db = g.to_device(b)
    g.rotate_add(L, 2, da, 2, db, out)
    got = out.download((2, 2, L, N))
    for r in range(2):
        assert np.array_equal(got[r], o.add(b[r], o.apply_galois(a[r], o.galois_elt(2), keys[o.galois_elt(2)])))
    g.rotate_add(L, 2, da, 4, db, db)
    got = db.download((2, 2, L, N))
    for r in range(2):
        assert np.array_equal(got[r], o.add(b[r], o.apply_galois(a[r], o.galois_elt(4), keys[o.galois_elt(4)])))
    # accumulateBFV(count = 6) within a row: 3 row rotations (seal_context.cpp:296-304)
    acc = g.to_device(a)
    tmp = g.alloc(2 * 2 * L * N)
    g.accumulate(L, 2, acc, 6, tmp)
    got = acc.download((2, 2, L, N))
    for r in range(2):
        t = a[r]
        for i in range(3):
            e = o.galois_elt(1 << i)
            t = o.add(t, o.apply_galois(t, e, keys[e]))
        assert np.array_equal(got[r], t)


def test_bfv_semantic_end_to_end(be, oracle):
    """Real keys: Dec(relin(mul(Enc x, Enc y))) = x*y slot-wise, computed on the GPU, decrypted by the oracle."""
    N, bits = 4096, [60, 40, 40, 60]
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=False)
    codec = oracle.BatchCodec(N, o.t)
    sk = o.keygen_secret(1)
    pk = o.keygen_public(sk, 2)
    rk = o.keygen_relin(sk, 3)
    rng = np.random.default_rng(5)
    x, y = rng.integers(-300, 300, N), rng.integers(-300, 300, N)
    cx, cy = o.encrypt(pk, codec.encode(x), 11), o.encrypt(pk, codec.encode(y), 12)
    L = g.L
    g.set_relin_key(rk)
    c3 = g.alloc(3 * L * N)
    g.bfv_multiply(L, 1, g.to_device(cx[None]), g.to_device(cy[None]), be.Context.pairwise(), c3)
    c2 = g.alloc(2 * L * N)
    g.relinearize(L, 1, c3, c2)
    ct = c2.download((2, L, N))
    v = codec.decode(o.bfv_decode_phase(o.decrypt_phase(ct, sk)))
    t = o.t
    v = np.where(v > t // 2, v - t, v)
    want = (x * y) % t
    want = np.where(want > t // 2, want - t, want)
    assert np.array_equal(v, want)
    g.close()


@pytest.mark.parametrize("bits", [[40], [60]])
def test_cfg1_literal_bfv_add_n4096_single_modulus(be, oracle, bits):
    """BASELINE configs[0] as worded: BFV EltwiseAdd, poly_modulus_degree 4096, a single coefficient modulus, batch 1.
    (The reference's own parameter rule cannot build it -- it always emits {60, b.., 60}, SURVEY 0.5 -- but the C ABI can:
    one prime is both the key level and the data level, no special prime, 128-bit security gate on.)  Evaluator::add
    (bfv eltwise .cpp:322) bit-exact against the oracle, and Enc -> add -> Dec = the cleartext sum."""
    N = 4096
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=True, device=0)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=True)
    assert g.moduli == o.moduli and g.t == o.t == 1032193 and g.L == 1 and g.K == 1
    rng = np.random.default_rng(bits[0])
    a, b = o.random_poly(rng, 1, 2), o.random_poly(rng, 1, 2)
    out = g.alloc(2 * N)
    g.add(1, 2, 1, g.to_device(a[None]), g.to_device(b[None]), be.Context.outer(0, 1, 0, 1), out)  # batch 1 x 1
    assert np.array_equal(out.download((2, 1, N)), o.add(a, b))
    # semantic: real keys, device-side encryption and decryption
    codec = oracle.BatchCodec(N, o.t)
    sk = o.keygen_secret(1)
    pk = o.keygen_public(sk, 2)
    g.set_public_key(pk)
    g.set_secret_key(sk)
    x, y = rng.integers(-5000, 5000, N), rng.integers(-5000, 5000, N)
    plains = np.stack([codec.encode(x), codec.encode(y)])
    cts, s, dec = g.alloc(2 * 2 * N), g.alloc(2 * N), g.alloc(N)
    g.encrypt(2, g.to_device(plains), 99, 0, cts)
    g.add(1, 2, 1, cts, cts, be.Context.outer(0, 1, 1, 1), s)
    g.decrypt(1, 2, 1, s, dec)
    v = codec.decode(dec.download((N,)))
    t = o.t
    v = np.where(v > t // 2, v - t, v)
    assert np.array_equal(v, x + y)
    # and the oracle decrypts the device's sum to the same plaintext
    ct = s.download((2, 1, N))
    assert np.array_equal(o.bfv_decode_phase(o.decrypt_phase(ct, sk)), dec.download((N,)))
    g.close()


@pytest.mark.parametrize("seed", range(6))
def test_bfv_random_parameter_chains(be, oracle, seed):
    """Randomly drawn BFV chains (ring size, 2..5 key primes of 35..60 bits in any order, plain modulus 16..22 bits): BEHZ
    multiply, relinearize and a row rotation, bit-exact against the oracle — both arithmetic engines in every role."""
    rng = np.random.default_rng(5000 + seed)
    N = int(rng.choice([1024, 2048, 4096]))
    K = int(rng.integers(2, 6))
    bits = [int(b) for b in rng.integers(35, 61, K)]
    pb = int(rng.integers(16, 23))
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False)
    assert g.moduli == o.moduli and g.t == o.t
    L = g.L
    n = int(rng.integers(1, 6))
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
    da, db = g.to_device(a), g.to_device(b)
    c3 = g.alloc(n * 3 * L * N)
    g.bfv_multiply(L, n, da, db, be.Context.pairwise(), c3)
    got3 = c3.download((n, 3, L, N))
    for r in range(n):
        assert np.array_equal(got3[r], o.bfv_multiply(a[r], b[r])), (bits, pb, N, r)
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    out = g.alloc(n * 2 * L * N)
    g.relinearize(L, n, c3, out)
    got = out.download((n, 2, L, N))
    for r in range(n):
        assert np.array_equal(got[r], o.relinearize(got3[r], rk)), (bits, pb, N, r)
    elt = g.galois_elt(int(rng.choice([1, 2, -1])))
    gk = o.random_kswitch_key(rng)
    g.set_galois_key(elt, gk)
    rot = g.alloc(n * 2 * L * N)
    g.apply_galois(L, n, da, elt, rot)
    gotr = rot.download((n, 2, L, N))
    for r in range(n):
        assert np.array_equal(gotr[r], o.apply_galois(a[r], elt, gk)), (bits, pb, N, r)
    g.close()


@pytest.mark.parametrize("name,N,bits,rows,cols,inner,layout", [
    ("default_shape", 4096, [60, 40, 40, 60], 3, 2, 4, "cba"),     # M0 column-major, M1 row-major: the bridge's layout
    ("row_major_both", 2048, [50, 40, 50], 2, 3, 5, "row"),        # other strides
    ("several_passes", 1024, [50, 40, 50], 2, 3, 700, "cba"),      # more than 4096 / (rows * cols) inner indices: two passes over k
])
def test_bfv_multiply_relin_accumulate_equals_the_reference_loop(be, oracle, name, N, bits, rows, cols, inner, layout):
    """he355_bfv_multiply_relin_accumulate against the loop it replaces (bfv cipherbatchaxis .cpp:398-410): for every (i, j), multiply,
    relinearize_inplace, add_inplace over the inner index -- here with the inner index inside the batch."""
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=False)
    assert g.moduli == o.moduli
    rng = np.random.default_rng(rows * 100 + inner)
    L = g.L
    distinct = min(inner, 6)  # the oracle's work: distinct products only; the long run repeats them
    a = np.stack([o.random_poly(rng, L, 2) for _ in range(rows * distinct)])
    b = np.stack([o.random_poly(rng, L, 2) for _ in range(distinct * cols)])
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    if layout == "cba":   # a(i, k) at k * rows + i, b(k, j) at k * cols + j
        ai = lambda i, k: (k % distinct) * rows + i
        bi = lambda k, j: (k % distinct) * cols + j
        a_si, a_sk, b_sk, b_sj = 1, rows, cols, 1
    else:                 # a(i, k) at i * inner + k, b(k, j) at k * cols + j
        ai = lambda i, k: i * distinct + (k % distinct)
        bi = lambda k, j: (k % distinct) * cols + j
        a_si, a_sk, b_sk, b_sj = inner, 1, cols, 1
    # device operands laid out for the full inner range
    A = np.zeros((rows * inner, 2, L, N), dtype=np.uint64)
    B = np.zeros((inner * cols, 2, L, N), dtype=np.uint64)
    for k in range(inner):
        for i in range(rows):
            A[i * a_si + k * a_sk] = a[ai(i, k)]
        for j in range(cols):
            B[k * b_sk + j * b_sj] = b[bi(k, j)]
    out = g.alloc(rows * cols * 2 * L * N)
    g.bfv_multiply_relin_accumulate(L, rows, cols, inner, g.to_device(A), a_si, a_sk, g.to_device(B), b_sk, b_sj, out)
    got = out.download((rows * cols, 2, L, N))
    mods = np.array(o.moduli[:L], dtype=object)
    for i in range(rows):
        for j in range(cols):
            terms = [o.relinearize(o.bfv_multiply(a[ai(i, k)], b[bi(k, j)]), rk) for k in range(distinct)]
            acc = np.zeros((2, L, N), dtype=object)
            for k in range(inner):
                acc = acc + terms[k % distinct].astype(object)
            want = np.stack([[acc[p][l] % int(mods[l]) for l in range(L)] for p in range(2)]).astype(np.uint64)
            assert np.array_equal(got[i * cols + j], want), (name, i, j)
    g.close()


def test_bfv_multiply_rejects_outputs_that_overlap_an_operand(be, oracle):
    """A chunk's products are written before the next chunk's operands are read (and the matrix-product entry reads its operands once
    per pass over the inner index): `out` inside an operand slab is refused with an invalid-argument error instead of corrupting it."""
    import ctypes as C

    class View:
        def __init__(self, buf, off):
            self.ptr = C.c_void_p(buf.ptr.value + off * 8)

    N, bits = 1024, [50, 40, 50]
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=False)
    rng = np.random.default_rng(3)
    L = g.L
    per = 2 * L * N
    g.set_relin_key(o.random_kswitch_key(rng))
    a = g.to_device(np.stack([o.random_poly(rng, L, 2) for _ in range(4)]))
    big = g.alloc(4 * per + 4 * 3 * L * N)  # operands first, room for the products behind them
    a.copy_into(big, 0, 4 * per)
    inside = View(big, per)
    behind = View(big, 4 * per)
    for call in (lambda: g.bfv_multiply(L, 4, big, a, be.Context.pairwise(), inside), lambda: g.bfv_multiply(L, 4, a, big, be.Context.outer(0, 2, 0, 2), inside),
                 lambda: g.bfv_multiply_relin_accumulate(L, 2, 1, 2, big, 1, 2, a, 1, 1, inside)):
        with pytest.raises(be.HE355Error) as ei:
            call()
        assert ei.value.code == be.E_INVALID_ARGS
    g.bfv_multiply(L, 4, big, a, be.Context.pairwise(), behind)  # disjoint ranges inside one allocation are fine
    g.sync()
    g.close()


def _residues(o, values, L):
    """Integer coefficient vector(s) -> residues [.., L, N] under the first L moduli."""
    return np.stack([np.array([int(v) % q for v in values], dtype=np.uint64) for q in o.moduli[:L]])


@pytest.mark.parametrize("base", ["device", "seal"])
@pytest.mark.parametrize("name,N,bits,pb", [
    ("default_d3", 4096, [60, 40, 40, 60], 20),          # <4, 6> instantiation of the coefficient kernels
    ("six_data_primes", 2048, [50, 40, 40, 45, 40, 60, 60], 22),   # <16, 24>, both engines among the data primes
    ("all_60_bit", 2048, [60, 60, 60, 60, 60], 31),      # the most auxiliary primes per data prime, the largest plain modulus
])
def test_bfv_multiply_is_independent_of_the_auxiliary_base(be, oracle, monkeypatch, base, name, N, bits, pb):
    """The BEHZ product is fixed by the base q and m_tilde alone (he_params.h, Params::aux): the device's 46-bit auxiliary primes
    (fp64 engine) and SEAL's 61-bit ones (HE355_BEHZ_BASE=seal) must both give the oracle's bits -- the oracle restates SEAL's
    RNSTool with SEAL's base -- on random operands AND on operands that drive every bound of the Shenoy-Kumaresan step to its
    extreme: all coefficients +-Q/2 with the signs that make the negacyclic sums of coefficient N-1 (no wrapped terms) and of
    coefficient 0 (all but one term wrapped) as large as they get, and all coefficients Q-1 / 0 / 1."""
    if base == "seal":
        monkeypatch.setenv("HE355_BEHZ_BASE", "seal")
    else:
        monkeypatch.delenv("HE355_BEHZ_BASE", raising=False)
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False)
    assert g.moduli == o.moduli and g.t == o.t
    L = g.L
    Q = 1
    for q in o.moduli[:L]:
        Q *= int(q)
    h = Q // 2
    alt = [h if i % 2 == 0 else h + 1 for i in range(N)]          # +Q/2, -Q/2 alternating
    first_neg = [h + 1] + [h] * (N - 1)                            # coefficient 0: -a_0 b_0 ... all wrapped terms add up
    polys = {
        "plus_half": _residues(o, [h] * N, L), "minus_half": _residues(o, [h + 1] * N, L), "alternating": _residues(o, alt, L),
        "first_negative": _residues(o, first_neg, L), "q_minus_1": _residues(o, [Q - 1] * N, L), "zero": _residues(o, [0] * N, L),
        "one": _residues(o, [1] * N, L),
    }
    rng = np.random.default_rng(77)
    cts = [np.stack([polys[x], polys[y]]) for x, y in [("plus_half", "plus_half"), ("minus_half", "plus_half"), ("alternating", "alternating"),
                                                         ("first_negative", "minus_half"), ("q_minus_1", "q_minus_1"), ("zero", "one"),
                                                         ("q_minus_1", "plus_half")]]
    cts += [o.random_poly(rng, L, 2) for _ in range(2)]
    a = np.stack(cts)
    n = len(cts)
    out = g.alloc(n * n * 3 * L * N)
    da = g.to_device(a)
    g.bfv_multiply(L, n * n, da, da, be.Context.outer(0, n, 0, n), out)
    got = out.download((n * n, 3, L, N))
    for i in range(n):
        for x in range(n):
            assert np.array_equal(got[i * n + x], o.bfv_multiply(a[i], a[x])), (base, name, i, x)
    g.close()


@pytest.mark.parametrize("walk", ["by_node", "by_level", "by_level_chunked"])
def test_rotate_sum_shares_naf_prefixes_bit_for_bit(pair, be, walk):
    """(walk: the trie node by node -- what batches of at most he355_set_latency_max ciphertexts take -- or level by level, all nodes of
    a level in one grouped kernel sequence in the NTT domain; `chunked`: with a chunk size that cuts through the groups.)
    he355_rotate_sum: out = in + sum_j rotate_rows(in, j * spacers), the inner loop of MatMultRow (bfv row .cpp:519-531), with every
    distinct NAF prefix key-switched once.  It must equal the reference's unshared loop -- one Evaluator::rotate_internal per step,
    each from `in` -- bit for bit (the oracle's rotate), and issue fewer key switches: for steps j * 2^k, j = 1 .. 15, the trie has 15
    nodes while the loop runs 26 key switches."""
    g, o, rng = pair
    L, N = g.L, g.N
    keys = {}
    k = 0
    while (1 << k) < N // 2:  # the default Galois key set of the reference: +-2^k row rotations
        for s in (1 << k, -(1 << k)):
            e = o.galois_elt(s)
            keys[e] = o.random_kswitch_key(rng)
            g.set_galois_key(e, keys[e])
        k += 1
    g.set_level_walk(walk != "by_node")
    g.set_latency_max(0)  # (two ciphertexts would otherwise stay within the latency shape, which is walked node by node)
    g.set_chunk(5 if walk == "by_level_chunked" else 1024)
    a = rand_cts(o, rng, 2, L)
    da = g.to_device(a)
    out = g.alloc(2 * 2 * L * N)
    spacers = (N // 2) // 16
    steps = [j * spacers for j in range(1, 16)]
    issued = g.rotate_sum(L, 2, da, steps, out)
    got = out.download((2, 2, L, N))
    unshared = 0
    for r in range(2):
        want = a[r].copy()
        for s in steps:
            want = o.add(want, o.rotate(a[r], s, keys))
        assert np.array_equal(got[r], want), r
    prefixes = set()
    for s in steps:  # the unshared loop (Evaluator::rotate_internal): the step's own key if present, else one key switch per NAF term
        if o.galois_elt(s) in keys:  # (e.g. 12 * 32 = 384 is the rotation by -128 of a 512-slot row: same Galois element, same key)
            terms = [s]
        else:
            v, i, terms = s, 0, []
            while v:
                zi = 2 - (v & 3) if v & 1 else 0
                v = (v - zi) >> 1
                if zi and abs(zi << i) != N // 2:
                    terms.append(zi * (1 << i))
                i += 1
        unshared += len(terms)
        elts = tuple(o.galois_elt(t) for t in terms)
        prefixes |= {elts[:k] for k in range(1, len(elts) + 1)}
    assert issued == len(prefixes) and issued < unshared, (issued, len(prefixes), unshared)
    # duplicates, a zero step and a negative step: still the plain sum
    steps2 = [3 * spacers, 3 * spacers, 0, -5 * spacers]
    g.rotate_sum(L, 2, da, steps2, out)
    got = out.download((2, 2, L, N))
    for r in range(2):
        want = o.add(a[r], a[r])
        for s in (3 * spacers, 3 * spacers, -5 * spacers):
            want = o.add(want, o.rotate(a[r], s, keys))
        assert np.array_equal(got[r], want), r
    g.set_level_walk(True)  # the context is shared by the module's tests: back to the defaults
    g.set_latency_max(None)
    g.set_chunk(1024)


@pytest.mark.parametrize("scheme", ["bfv", "ckks"])
def test_rotate_sum_level_sum_inside_the_key_switch(be, oracle, scheme):
    """From batches whose grid fills the chip (eight-ciphertext groups x tiles >= 512 blocks) he355_rotate_sum's level walk lets the fused
    k_k3 add every node's ciphertext into the sum itself -- one block per (tile, eight ciphertexts) walking the level's nodes, nodes
    nothing starts from never written (KsGroups::sum_out) -- instead of k_sum_groups' pass over them.  Same bits as that pass (forced by a
    chunk below the batch, which rules the in-kernel sum out) and as the reference's unshared loop on sampled ciphertexts; steps with
    repeats (a node several steps end at), inner nodes that are ends themselves, a zero step."""
    N, bits = 8192, [60, 40, 40, 60]
    n = 176  # 3 primes x 8 rows x 22 eight-ciphertext groups = 528 blocks
    sch_g, sch_o = (be.SCHEME_BFV, oracle.SCHEME_BFV) if scheme == "bfv" else (be.SCHEME_CKKS, oracle.SCHEME_CKKS)
    kw = dict(plain_bits=20) if scheme == "bfv" else {}
    g = be.Context(sch_g, N, bit_sizes=bits, sec128=False, device=0, **kw)
    o = oracle.Context(sch_o, N, bit_sizes=bits, sec128=False, **kw)
    rng = np.random.default_rng(77)
    L = g.L
    try:
        keys = {}
        k = 0
        while (1 << k) < N // 2:
            for s in (1 << k, -(1 << k)):
                e = o.galois_elt(s)
                keys[e] = o.random_kswitch_key(rng)
                g.set_galois_key(e, keys[e])
            k += 1
        a = rand_cts(o, rng, n, L)
        da = g.to_device(a)
        steps = [1, 2, 3, 3, 5, 6, 7, 0, 1, 11, -3, 96]
        out, out2 = g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
        g.set_latency_max(0)
        issued = g.rotate_sum(L, n, da, steps, out)
        got = out.download((n, 2, L, N))
        g.set_chunk(n - 8)  # a launch no longer holds whole groups: every level through k_sum_groups
        assert g.rotate_sum(L, n, da, steps, out2) == issued
        assert np.array_equal(out2.download((n, 2, L, N)), got)
        for r in (0, 7, 8, 100, n - 1):
            want = a[r].copy()
            for s in steps:
                want = o.add(want, o.rotate(a[r], s, keys) if s else a[r])
            assert np.array_equal(got[r], want), r
    finally:
        g.close()
