"""GPU parity tests (run with -m gpu on an MI355X): every evaluator op of the C ABI, bit-exact against the
CPU oracle on the same seeded inputs, on ring sizes the oracle finishes in seconds and — through sampled
results and size-independent properties — at BASELINE.json's full sizes.  Integer work: the bar is
bit-exact equality (np.array_equal), no tolerance anywhere."""
import importlib
import os
import sys
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    mod = importlib.import_module("reference-seal-backend_amd")
    if mod.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run on the MI355X box")
    return mod


CONFIGS = {
    # name: (N, key-level bit sizes, force_u64)
    "n1024_mixed": (1024, [50, 40, 40, 50], False),
    "n2048_two_primes": (2048, [45, 52], False),  # one data prime + the special prime: the shortest chain that can key-switch
    "n2048_f64": (2048, [46, 40, 40, 46], False),
    "n4096_u64_forced": (4096, [60, 45, 45, 60], True),
    "n8192_default": (8192, [60, 45, 60], False),
    "n16384_d4": (16384, [60, 45, 45, 45, 60], False),
    "n32768_d4": (32768, [60, 45, 45, 45, 60], False),
}


def make_pair(be, oracle, name):
    N, bits, force = CONFIGS[name]
    if force:
        os.environ["HE355_FORCE_U64"] = "1"
    try:
        g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
    finally:
        os.environ.pop("HE355_FORCE_U64", None)
    o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    assert g.moduli == o.moduli
    if force:
        assert not any(g.fp64)
    return g, o


@pytest.fixture(scope="module", params=list(CONFIGS))
def pair(request, be, oracle):
    g, o = make_pair(be, oracle, request.param)
    rng = np.random.default_rng(zlib.crc32(request.param.encode()))
    yield g, o, rng
    g.close()


def rand_cts(o, rng, n, L, size=2):
    return np.stack([o.random_poly(rng, L, size) for _ in range(n)])  # [n, size, L, N]


def test_ntt_roundtrip_and_oracle(pair):
    g, o, rng = pair
    K, N = g.K, g.N
    polys = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in g.moduli]) for _ in range(3)])  # [3,K,N]
    d = g.to_device(polys)
    g.ntt(d, 3 * K, list(range(K)))
    f = d.download(polys.shape)
    for it in range(3):
        for i in range(K):
            assert np.array_equal(f[it, i], o.ntt(i, polys[it, i])), (it, i)
    g.ntt(d, 3 * K, list(range(K)), inverse=True)
    assert np.array_equal(d.download(polys.shape), polys)


def test_add_outer_product(pair, be):
    g, o, rng = pair
    L = g.L
    a, b = rand_cts(o, rng, 2, L), rand_cts(o, rng, 3, L)
    da, db = g.to_device(a), g.to_device(b)
    out = g.alloc(6 * 2 * L * g.N)
    g.add(L, 2, 6, da, db, be.Context.outer(0, 2, 0, 3), out)
    got = out.download((6, 2, L, g.N))
    for i in range(2):
        for x in range(3):
            assert np.array_equal(got[i * 3 + x], o.add(a[i], b[x]))  # r = i*b1 + x (ckks eltwise .cpp:336)
    g.add(L, 2, 2, da, db, be.Context.pairwise(0, 1), out, sub=True)
    got = out.download((6, 2, L, g.N))
    for r in range(2):
        want = np.empty_like(a[r])
        import oracle as ho
        ho.lib().ho_sub(o.h, L, 2, ho._p(a[r]), ho._p(b[r + 1]), ho._p(want))
        assert np.array_equal(got[r], want)


def test_multiply_size3(pair, be):
    g, o, rng = pair
    L = g.L
    a, b = rand_cts(o, rng, 3, L), rand_cts(o, rng, 2, L)
    da, db = g.to_device(a), g.to_device(b)
    out = g.alloc(6 * 3 * L * g.N)
    g.multiply(L, 6, da, db, be.Context.outer(0, 3, 0, 2), out)
    got = out.download((6, 3, L, g.N))
    for i in range(3):
        for x in range(2):
            assert np.array_equal(got[i * 2 + x], o.multiply_ntt(a[i], b[x]))


@pytest.mark.parametrize("drop", [0, 1])
def test_multiply_relin_and_rescale(pair, be, drop):
    g, o, rng = pair
    L = g.L - drop
    if L < 2:
        pytest.skip("needs two data residues")
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    n = 5  # not a multiple of the wave count per block
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
    da, db = g.to_device(a), g.to_device(b)
    out = g.alloc(n * 2 * L * g.N)
    g.set_chunk(2)  # exercise chunking with a ragged tail
    g.multiply_relin(L, n, da, db, be.Context.pairwise(), out)
    got = out.download((n, 2, L, g.N))
    want = [o.relinearize(o.multiply_ntt(a[r], b[r]), rk) for r in range(n)]
    for r in range(n):
        assert np.array_equal(got[r], want[r]), r
    out2 = g.alloc(n * 2 * (L - 1) * g.N)
    g.set_chunk(32)
    g.multiply_relin(L, n, da, db, be.Context.pairwise(), out2, rescale=True)
    got2 = out2.download((n, 2, L - 1, g.N))
    for r in range(n):
        assert np.array_equal(got2[r], o.rescale(want[r])), r


def test_relinearize_and_rescale_standalone(pair, be):
    g, o, rng = pair
    L = g.L
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    ct3 = rand_cts(o, rng, 3, L, size=3)
    d3 = g.to_device(ct3)
    out = g.alloc(3 * 2 * L * g.N)
    g.relinearize(L, 3, d3, out)
    got = out.download((3, 2, L, g.N))
    for r in range(3):
        assert np.array_equal(got[r], o.relinearize(ct3[r], rk))
    for size in (2, 3) if L >= 2 else ():  # nothing to rescale to on a one-prime level
        src = rand_cts(o, rng, 3, L, size=size)
        ds = g.to_device(src)
        o2 = g.alloc(3 * size * (L - 1) * g.N)
        g.rescale(L, size, 3, ds, o2)
        got = o2.download((3, size, L - 1, g.N))
        for r in range(3):
            assert np.array_equal(got[r], o.rescale(src[r])), (size, r)


def test_multiply_accumulate_and_relinearize_rescale(pair, be):
    """The CipherBatchAxis inner loop (multiply, then multiply + add_inplace over the inner dimension; ckks
    cipherbatchaxis .cpp:404-420) and its relinearize + rescale tail (:436-437), with strided operand layouts."""
    g, o, rng = pair
    L = g.L
    r0, c0, c1 = 3, 4, 2
    a, b = rand_cts(o, rng, c0 * r0, L), rand_cts(o, rng, c0 * c1, L)  # a column-major [k][i], b row-major [k][j]
    da, db = g.to_device(a), g.to_device(b)
    d3 = g.alloc(r0 * c1 * 3 * L * g.N)
    g.multiply_accumulate(L, r0, c1, c0, da, 1, r0, db, c1, 1, d3)
    got = d3.download((r0 * c1, 3, L, g.N))
    want = []
    for i in range(r0):
        for j in range(c1):
            acc = o.multiply_ntt(a[0 * r0 + i], b[0 * c1 + j])
            for k in range(1, c0):
                acc = o.add(acc, o.multiply_ntt(a[k * r0 + i], b[k * c1 + j]))
            want.append(acc)
            assert np.array_equal(got[i * c1 + j], acc), (i, j)
    # the transposed layouts give the same sums: a row-major [i][k], b column-major [j][k]
    at = np.ascontiguousarray(a.reshape(c0, r0, *a.shape[1:]).swapaxes(0, 1)).reshape(a.shape)
    bt = np.ascontiguousarray(b.reshape(c0, c1, *b.shape[1:]).swapaxes(0, 1)).reshape(b.shape)
    d3t = g.alloc(r0 * c1 * 3 * L * g.N)
    g.multiply_accumulate(L, r0, c1, c0, g.to_device(at), c0, 1, g.to_device(bt), 1, c0, d3t)
    assert np.array_equal(d3t.download(got.shape), got)
    if L >= 2:
        rk = o.random_kswitch_key(rng)
        g.set_relin_key(rk)
        out = g.alloc(r0 * c1 * 2 * (L - 1) * g.N)
        g.set_chunk(4)  # ragged tail: 6 results
        g.relinearize_rescale(L, r0 * c1, d3, out)
        g.set_chunk(32)
        res = out.download((r0 * c1, 2, L - 1, g.N))
        for r in range(r0 * c1):
            assert np.array_equal(res[r], o.rescale(o.relinearize(want[r], rk))), r


def test_plain_ops_mod_switch_and_sum(pair, be):
    """multiply_plain / add_plain with per-op and broadcast plaintexts, CKKS mod_switch_to (drop residues) and the
    collapse sum (seal_context.cpp:389-401, 451-454)."""
    g, o, rng = pair
    L, n = g.L, 5
    cts = rand_cts(o, rng, n, L)
    pts = np.stack([o.random_poly(rng, L, 1)[0] for _ in range(n)])  # [n, L, N] NTT-form plaintexts
    dc, dp = g.to_device(cts), g.to_device(pts)
    out = g.alloc(n * 2 * L * g.N)
    g.multiply_plain(L, 2, n, dc, dp, be.Context.pairwise(), out)
    got = out.download(cts.shape)
    for r in range(n):
        assert np.array_equal(got[r], o.multiply_plain(cts[r], pts[r])), r
    g.multiply_plain(L, 2, n, dc, dp, be.Context.outer(0, n, 2, 1), out)  # one plaintext (index 2) for all
    got = out.download(cts.shape)
    for r in range(n):
        assert np.array_equal(got[r], o.multiply_plain(cts[r], pts[2])), r
    g.add_plain(L, 2, n, dc, dp, be.Context.pairwise(), out)
    got = out.download(cts.shape)
    for r in range(n):
        assert np.array_equal(got[r], o.add_plain(cts[r], pts[r])), r
    ct3 = rand_cts(o, rng, 2, L, size=3)
    d3, o3 = g.to_device(ct3), g.alloc(2 * 3 * L * g.N)
    g.multiply_plain(L, 3, 2, d3, dp, be.Context.pairwise(), o3)
    assert np.array_equal(o3.download(ct3.shape)[1], o.multiply_plain(ct3[1], pts[1]))
    # sum of the n ciphertexts
    ds = g.alloc(2 * L * g.N)
    g.sum(L, 2, n, dc, ds)
    want = cts[0]
    for r in range(1, n):
        want = o.add(want, cts[r])
    assert np.array_equal(ds.download(want.shape), want)
    if L >= 2:
        for L_to in range(1, L + 1):
            dd = g.alloc(n * 2 * L_to * g.N)
            g.mod_switch_drop(L, L_to, n * 2, dc, dd)
            got = dd.download((n, 2, L_to, g.N))
            for r in range(n):
                assert np.array_equal(got[r], o.mod_switch_drop(cts[r], L_to)), (L_to, r)
        dd = g.alloc(n * (L - 1) * g.N)
        g.mod_switch_drop(L, L - 1, n, dp, dd)  # plaintexts: one polynomial each
        assert np.array_equal(dd.download((n, L - 1, g.N)), pts[:, : L - 1])


def test_galois_rotate_accumulate(pair, be):
    g, o, rng = pair
    L, N = g.L, g.N
    steps = [1, 2, 4, -1]
    keys = {}
    for s in steps:
        elt = o.galois_elt(s)
        keys[elt] = o.random_kswitch_key(rng)
        g.set_galois_key(elt, keys[elt])
    conj = 2 * N - 1
    keys[conj] = o.random_kswitch_key(rng)
    g.set_galois_key(conj, keys[conj])
    a = rand_cts(o, rng, 3, L)
    da = g.to_device(a)
    out = g.alloc(3 * 2 * L * N)
    for elt in (o.galois_elt(1), conj):
        g.apply_galois(L, 3, da, elt, out)
        got = out.download((3, 2, L, N))
        for r in range(3):
            assert np.array_equal(got[r], o.apply_galois(a[r], elt, keys[elt])), (elt, r)
    # rotate by 3: no key for 3^3 -> NAF 3 = -1 + 4, applied in that order (Evaluator::rotate_internal)
    g.rotate(L, 3, da, 3, out)
    got = out.download((3, 2, L, N))
    for r in range(3):
        t = o.apply_galois(a[r], o.galois_elt(-1), keys[o.galois_elt(-1)])
        t = o.apply_galois(t, o.galois_elt(4), keys[o.galois_elt(4)])
        assert np.array_equal(got[r], t), r
    # accumulateCKKS(count=5): rotations = bit_count(5) = 3 (seal_context.cpp:331-339)
    acc = g.to_device(a)
    tmp = g.alloc(3 * 2 * L * N)
    g.accumulate(L, 3, acc, 5, tmp)
    got = acc.download((3, 2, L, N))
    for r in range(3):
        t = a[r]
        for i in range(3):
            e = o.galois_elt(1 << i)
            t = o.add(t, o.apply_galois(t, e, keys[e]))
        assert np.array_equal(got[r], t), r
    # rotate_add: out = addend + rotate(in), the add folded into the rotation's first kernel
    b = rand_cts(o, rng, 3, L)
    db = g.to_device(b)
    k1 = keys[o.galois_elt(1)]
    g.rotate_add(L, 3, da, 1, db, out)
    got = out.download((3, 2, L, N))
    for r in range(3):
        assert np.array_equal(got[r], o.add(b[r], o.apply_galois(a[r], o.galois_elt(1), k1))), r
    g.rotate_add(L, 3, da, 1, db, db)  # add in place: the addend is the output
    got = db.download((3, 2, L, N))
    for r in range(3):
        assert np.array_equal(got[r], o.add(b[r], o.apply_galois(a[r], o.galois_elt(1), k1))), r
    db = g.to_device(b)
    g.rotate_add(L, 3, da, 3, db, out)  # NAF 3 = -1 + 4: the addend joins the last step
    got = out.download((3, 2, L, N))
    for r in range(3):
        t = o.apply_galois(a[r], o.galois_elt(-1), keys[o.galois_elt(-1)])
        t = o.apply_galois(t, o.galois_elt(4), keys[o.galois_elt(4)])
        assert np.array_equal(got[r], o.add(b[r], t)), r
    g.rotate_add(L, 3, da, 0, db, out)  # step 0: plain add
    got = out.download((3, 2, L, N))
    for r in range(3):
        assert np.array_equal(got[r], o.add(b[r], a[r])), r
    with pytest.raises(be.HE355Error):
        g.rotate_add(L, 3, da, 3, db, db)  # several Galois steps cannot add in place
    with pytest.raises(be.HE355Error):
        g.rotate(L, 3, da, 8, out)  # power-of-two step without a key: "Galois key not present"


def test_block_sync_debug_mode_agrees(be, oracle):
    """The wavefront-scope LDS hand-off and the workgroup-barrier variant give identical results."""
    os.environ["HE355_BLOCK_SYNC"] = "1"
    try:
        g, o = make_pair(be, oracle, "n2048_f64")
    finally:
        os.environ.pop("HE355_BLOCK_SYNC", None)
    rng = np.random.default_rng(77)
    L = g.L
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    a, b = rand_cts(o, rng, 3, L), rand_cts(o, rng, 3, L)
    out = g.alloc(3 * 2 * (L - 1) * g.N)
    g.multiply_relin(L, 3, g.to_device(a), g.to_device(b), be.Context.pairwise(), out, rescale=True)
    got = out.download((3, 2, L - 1, g.N))
    for r in range(3):
        assert np.array_equal(got[r], o.rescale(o.relinearize(o.multiply_ntt(a[r], b[r]), rk)))
    g.close()
    # restore the default mode for later contexts
    g2, _ = make_pair(be, oracle, "n1024_mixed")
    g2.close()


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json full sizes
# ---------------------------------------------------------------------------------------------------------
def test_cfg2_full_size_multiply(be, oracle):
    """configs[1]: CKKS EltwiseMult, N=2^14, depth 8 (L=8), 256 results as a 16x16 outer product."""
    bits = be.chain_bits(8, 45)
    g = be.Context(be.SCHEME_CKKS, 16384, bit_sizes=bits, device=0)
    o = oracle.Context(oracle.SCHEME_CKKS, 16384, bit_sizes=bits)
    L, N = g.L, g.N
    assert L == 8
    pm = list(range(L))
    da, db = g.alloc(16 * 2 * L * N), g.alloc(16 * 2 * L * N)
    g.fill_uniform(da, 16 * 2 * L, pm, 1234)
    g.fill_uniform(db, 16 * 2 * L, pm, 4321)
    out = g.alloc(256 * 3 * L * N)
    g.multiply(L, 256, da, db, be.Context.outer(0, 16, 0, 16), out)
    a, b = da.download((16, 2, L, N)), db.download((16, 2, L, N))
    for q, poly in zip(g.moduli[:L], a[0, 0]):
        assert poly.max() < q
    got = out.download((256, 3, L, N))
    for r in (0, 17, 100, 255):
        assert np.array_equal(got[r], o.multiply_ntt(a[r // 16], b[r % 16])), r
    # size-independent property over ALL 256 results: the tensor is symmetric in its operands
    out_t = g.alloc(256 * 3 * L * N)
    g.multiply(L, 256, db, da, be.Context.outer(0, 16, 0, 16), out_t)
    got_t = out_t.download((256, 3, L, N)).reshape(16, 16, 3, L, N)
    assert np.array_equal(got.reshape(16, 16, 3, L, N), got_t.transpose(1, 0, 2, 3, 4))
    g.close()


def test_cfg3_full_size_mul_relin_rescale(be, oracle):
    """configs[2] (the headline): CKKS multiply -> relinearize -> rescale, N=2^15, depth 16, batch 1024 x 1.
    32 sampled results (chunk boundaries included) are checked bit-for-bit against the oracle; all 1024 are checked through a
    size-independent property (batch position does not matter: a permuted batch gives permuted results)."""
    bits = be.chain_bits(16, 45)
    g = be.Context(be.SCHEME_CKKS, 32768, bit_sizes=bits, device=0)
    o = oracle.Context(oracle.SCHEME_CKKS, 32768, bit_sizes=bits)
    L, N, n = g.L, g.N, 1024
    assert L == 16 and g.K == 17
    pm = list(range(L))
    da, db = g.alloc(n * 2 * L * N), g.alloc(1 * 2 * L * N)
    g.fill_uniform(da, n * 2 * L, pm, 1234)
    g.fill_uniform(db, 2 * L, pm, 99)
    g.set_relin_key_synthetic(7)
    # the oracle needs the same key: regenerate it on the host through a second device fill + download
    kbuf = g.alloc(L * 2 * g.K * N)
    g.fill_uniform(kbuf, L * 2 * g.K, list(range(g.K)), 7)
    rk = kbuf.download((L, 2, g.K, N))
    out = g.alloc(n * 2 * (L - 1) * N)
    g.multiply_relin(L, n, da, db, be.Context.outer(0, n, 0, 1), out, rescale=True)
    b = db.download((1, 2, L, N))[0]
    stride_in, stride_out = 2 * L * N, 2 * (L - 1) * N
    import ctypes as C
    # 32 results against the oracle (its multithreaded batch loop): the first and last rows, both sides of every boundary of the
    # 256-op chunks of the second pass below, and a spread in between.  (The default chunk is the whole batch: one chunk, one stream.)
    rows = sorted({0, 1, 2, 63, 127, 128, 200, 254, 255, 256, 257, 300, 383, 384, 500, 510, 511, 512, 513, 600, 639, 640, 700, 766, 767, 768, 769,
                   900, 1000, 1021, 1022, 1023})
    a_s = np.empty((len(rows), 2, L, N), dtype=np.uint64)
    got = np.empty((len(rows), 2, L - 1, N), dtype=np.uint64)
    for k, r in enumerate(rows):
        be._check(be.lib().he355_download(g.h, a_s[k].ctypes.data_as(C.c_void_p), C.c_void_p(da.ptr.value + r * stride_in * 8), a_s[k].nbytes))
        be._check(be.lib().he355_download(g.h, got[k].ctypes.data_as(C.c_void_p), C.c_void_p(out.ptr.value + r * stride_out * 8), got[k].nbytes))
    want = o.batch_outer(oracle.OP_MUL_RELIN_RESCALE, a_s, b[None], rk)
    for k, r in enumerate(rows):
        assert np.array_equal(got[k], want[k]), r
    # property over the whole batch: op r of a run that starts at value_index 512 equals op r+512 of the full run
    out_hi = g.alloc(512 * stride_out)
    g.multiply_relin(L, 512, da, db, be.Context.outer(512, 512, 0, 1), out_hi, rescale=True)
    full = out.download()
    hi = out_hi.download()
    assert np.array_equal(full[512 * stride_out:], hi)
    # the same batch cut into four chunks of 256 that alternate between the two streams (the schedule a smaller memory budget or
    # he355_set_chunk selects): every result identical to the one-chunk run, hence to the oracle on the sampled rows
    g.set_chunk(256)
    out_4 = g.alloc(n * stride_out)
    g.multiply_relin(L, n, da, db, be.Context.outer(0, n, 0, 1), out_4, rescale=True)
    assert np.array_equal(full, out_4.download())
    out_4.free()
    # and the result does not depend on the chunking of the batch
    g.set_chunk(7)
    out_c = g.alloc(64 * stride_out)
    g.multiply_relin(L, 64, da, db, be.Context.outer(0, 64, 0, 1), out_c, rescale=True)
    assert np.array_equal(full[: 64 * stride_out], out_c.download())
    g.close()


def test_cfg4_full_size_dot_product(be, oracle):
    """configs[3]: CKKS DotProduct, vector length 4096, N=2^15, L=16: multiply -> relinearize -> accumulateCKKS(4096)
    = 1 + 12 key switches per result (seal_context.cpp:331-339).  All eight results checked bit-for-bit against the oracle."""
    bits = be.chain_bits(16, 45)
    g = be.Context(be.SCHEME_CKKS, 32768, bit_sizes=bits, device=0)
    o = oracle.Context(oracle.SCHEME_CKKS, 32768, bit_sizes=bits)
    L, N, K, n = g.L, g.N, g.K, 8
    pm = list(range(L))
    da, db = g.alloc(n * 2 * L * N), g.alloc(2 * L * N)
    g.fill_uniform(da, n * 2 * L, pm, 11)
    g.fill_uniform(db, 2 * L, pm, 12)
    kbuf = g.alloc(L * 2 * K * N)

    def key(seed):
        g.fill_uniform(kbuf, L * 2 * K, list(range(K)), seed)
        return kbuf.download((L, 2, K, N))
    g.set_relin_key_synthetic(100)
    rk = key(100)
    gks = {}
    for i in range(12):
        e = o.galois_elt(1 << i)
        g.set_galois_key_synthetic(e, 200 + i)
        gks[e] = key(200 + i)
    out, tmp = g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
    g.multiply_relin(L, n, da, db, be.Context.outer(0, n, 0, 1), out)
    g.accumulate(L, n, out, 4096, tmp)
    a, b = da.download((n, 2, L, N)), db.download((1, 2, L, N))[0]
    got = out.download((n, 2, L, N))
    want = o.batch_outer(oracle.OP_DOT, a, b[None], rk, gks, 4096)  # all 8 results, the oracle's own operate() loop
    for r in range(n):
        assert np.array_equal(got[r], want[r]), r
    t = o.relinearize(o.multiply_ntt(a[0], b), rk)  # and the loop spelled out for one of them
    for i in range(12):
        e = o.galois_elt(1 << i)
        t = o.add(t, o.apply_galois(t, e, gks[e]))
    assert np.array_equal(got[0], t)
    g.close()


def test_empty_batches_and_argument_errors(be, oracle):
    """n = 0 is a no-op for every batched entry point; bad levels / sizes / missing keys are reported as HE355_E_INVALID_ARGS
    (the bridge maps them to the reference's error codes), never executed."""
    g = be.Context(be.SCHEME_CKKS, 2048, bit_sizes=[60, 40, 40, 60], sec128=False, device=0)
    L, N = g.L, g.N
    buf = g.alloc(4 * 3 * L * N)
    ix = be.Context.pairwise()
    g.add(L, 2, 0, buf, buf, ix, buf)
    g.multiply(L, 0, buf, buf, ix, buf)
    g.rescale(L, 2, 0, buf, buf)
    g.mod_switch_drop(L, L - 1, 0, buf, buf)
    g.multiply_plain(L, 2, 0, buf, buf, ix, buf)
    g.set_relin_key_synthetic(1)
    g.multiply_relin(L, 0, buf, buf, ix, buf, rescale=True)
    g.relinearize(L, 0, buf, buf)
    g.sync()
    for bad in (lambda: g.add(L + 1, 2, 1, buf, buf, ix, buf),            # level above the top
                lambda: g.add(0, 2, 1, buf, buf, ix, buf),                # level 0
                lambda: g.rescale(1, 2, 1, buf, buf),                     # nothing left to drop
                lambda: g.rescale(L, 4, 1, buf, buf),                     # size out of range
                lambda: g.mod_switch_drop(L, L + 1, 1, buf, buf),         # cannot switch upwards
                lambda: g.apply_galois(L, 1, buf, 4, buf),                # even Galois element
                lambda: g.rotate(L, 1, buf, 1, buf),                      # Galois key for step 1 not set
                lambda: g.sum(L, 2, 0, buf, buf),                         # nothing to sum
                lambda: g.decrypt(L, 2, 1, buf, buf),                     # secret key not set
                lambda: g.bfv_multiply(L, 1, buf, buf, ix, buf)):         # BFV op on a CKKS context
        with pytest.raises(be.HE355Error) as ei:
            bad()
        assert ei.value.code == 1, ei.value
    g.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_parameter_chains(be, oracle, seed):
    """Randomly drawn chains (ring size, number of primes, bit sizes 30..60 in any position, so both arithmetic engines end up
    anywhere in the chain, including an fp64-engine special prime): the fused multiply -> relinearize -> rescale pipeline, the
    unfused building blocks and a rotation, all bit-exact against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([1024, 2048, 4096, 8192]))
    K = int(rng.integers(2, 7))
    bits = [int(b) for b in rng.integers(30, 61, K)]
    g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    assert g.moduli == o.moduli
    L = g.L
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    n = int(rng.integers(1, 12))
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
    da, db = g.to_device(a), g.to_device(b)
    g.set_chunk(int(rng.integers(1, 9)))
    want = [o.relinearize(o.multiply_ntt(a[r], b[r]), rk) for r in range(n)]
    out = g.alloc(n * 2 * L * N)
    g.multiply_relin(L, n, da, db, be.Context.pairwise(), out)
    got = out.download((n, 2, L, N))
    for r in range(n):
        assert np.array_equal(got[r], want[r]), (bits, N, r)
    if L >= 2:
        out2 = g.alloc(n * 2 * (L - 1) * N)
        g.multiply_relin(L, n, da, db, be.Context.pairwise(), out2, rescale=True)
        got2 = out2.download((n, 2, L - 1, N))
        for r in range(n):
            assert np.array_equal(got2[r], o.rescale(want[r])), (bits, N, r)
    step = int(rng.integers(1, N // 2))
    # SEAL's NAF terms of the step, least significant first (Evaluator::rotate_internal); a term of N/2 is no rotation
    elt_list, x, i = [], step, 0
    while x:
        z = 2 - (x & 3) if x & 1 else 0
        x = (x - z) >> 1
        if z and (1 << i) != N // 2:
            elt_list.append(g.galois_elt(z * (1 << i)))
        i += 1
    keys = {}
    for e in set(elt_list):
        keys[e] = o.random_kswitch_key(rng)
        g.set_galois_key(e, keys[e])
    rot = g.alloc(n * 2 * L * N)
    g.rotate(L, n, da, step, rot)
    gotr = rot.download((n, 2, L, N))
    for r in range(min(n, 2)):
        w = a[r]
        for e in elt_list:
            w = o.apply_galois(w, e, keys[e])
        assert np.array_equal(gotr[r], w), (bits, N, step, r)
    g.close()


@pytest.mark.parametrize("N,bits", [(32768, [47, 46, 45, 44, 47]), (4096, [46, 47, 60, 46])])
def test_wide_fp64_engine_primes(be, oracle, N, bits):
    """46/47-bit primes still run on the fp64 engine (q < 2^47) but their lazy column-pass values no longer fit the 48-bit
    digit rows without re-centring (k_k2 re-centres for q >= 2^45): multiply -> relinearize (-> rescale) bit-exact, at the
    ring size with the most column stages and at a small one."""
    rng = np.random.default_rng(4747 + N)
    g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    assert g.moduli == o.moduli
    assert g.fp64 == [b <= 47 for b in bits]
    L, n = g.L, 3
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
    da, db = g.to_device(a), g.to_device(b)
    out = g.alloc(n * 2 * L * N)
    g.multiply_relin(L, n, da, db, be.Context.pairwise(), out)
    got = out.download((n, 2, L, N))
    want = [o.relinearize(o.multiply_ntt(a[r], b[r]), rk) for r in range(n)]
    for r in range(n):
        assert np.array_equal(got[r], want[r]), r
    out2 = g.alloc(n * 2 * (L - 1) * N)
    g.multiply_relin(L, n, da, db, be.Context.pairwise(), out2, rescale=True)
    got2 = out2.download((n, 2, L - 1, N))
    for r in range(n):
        assert np.array_equal(got2[r], o.rescale(want[r])), r
    g.close()


def _naf_terms(step, N):
    """SEAL's NAF terms of a rotation step, least significant first (Evaluator::rotate_internal); a term of N/2 is no rotation."""
    neg, x, i, out = step < 0, abs(step), 0, []
    while x:
        z = 2 - (x & 3) if x & 1 else 0
        x = (x - z) >> 1
        if z and (1 << i) != N // 2:
            out.append((-z if neg else z) * (1 << i))
        i += 1
    return out


@pytest.mark.parametrize("grouped", [True, False])
def test_rotate_each_batches_by_galois_element(pair, be, grouped):
    """(grouped: every ciphertext with a t-th NAF term in ONE key-switch sequence, each op with its own Galois element and key; else one
    sequence per Galois element.)  he355_rotate_each: ciphertext i rotated by its own step (collapseCKKS's rotate_vector(dot_i, -i) loop), key switches batched
    per Galois element: equal to rotating one ciphertext at a time, which is the oracle's apply_galois chain of the NAF terms."""
    g, o, rng = pair
    L, N = g.L, g.N
    keys = {}
    for k in range(4):
        for s in (1 << k, -(1 << k)):
            e = o.galois_elt(s)
            keys[e] = o.random_kswitch_key(rng)
            g.set_galois_key(e, keys[e])
    steps = [0, -1, -2, -3, -4, -5, -6, -7, 3, 5, 8, -10]
    n = len(steps)
    a = rand_cts(o, rng, n, L)
    da = g.to_device(a)
    out = g.alloc(n * 2 * L * N)
    g.set_level_walk(grouped)
    g.rotate_each(L, n, da, steps, out)
    g.set_level_walk(True)
    got = out.download((n, 2, L, N))
    for r, s in enumerate(steps):
        w = a[r]
        terms = [s] if s and o.galois_elt(s) in keys else _naf_terms(s, N)
        for t in terms:
            w = o.apply_galois(w, o.galois_elt(t), keys[o.galois_elt(t)])
        assert np.array_equal(got[r], w), (r, s)
    one = g.alloc(2 * L * N)
    g.rotate(L, 1, g.to_device(a[5:6]), steps[5], one)  # and equal to the single-ciphertext rotation
    assert np.array_equal(one.download((1, 2, L, N))[0], got[5])
    with pytest.raises(be.HE355Error):
        g.rotate_each(L, 2, da, [1, 16], out)  # 16: no key, a single NAF term
    with pytest.raises(be.HE355Error):
        g.rotate_each(L, n, da, steps, da)  # not in place


@pytest.mark.parametrize("walk", ["by_node", "by_level", "by_level_chunked"])
def test_rotate_sum_shares_naf_prefixes(pair, be, walk):
    """(walk: node by node -- batches of at most he355_set_latency_max ciphertexts -- or level by level with grouped key switches.)
    he355_rotate_sum (CKKS): out = in + sum_j rotate_vector(in, steps[j]) with every distinct NAF prefix key-switched once -- equal
    bit for bit to the reference's loop of independent rotations + add_inplace (ckks row .cpp:502-514), fewer key switches."""
    g, o, rng = pair
    L, N = g.L, g.N
    keys = {}
    for k in range(6):
        for s in (1 << k, -(1 << k)):
            e = o.galois_elt(s)
            keys[e] = o.random_kswitch_key(rng)
            g.set_galois_key(e, keys[e])
    steps = [2 * j for j in range(1, 12)]  # 2 .. 22: NAF terms within +-32
    g.set_level_walk(walk != "by_node")
    g.set_latency_max(0)  # (two ciphertexts would otherwise stay within the latency shape, which is walked node by node)
    g.set_chunk(5 if walk == "by_level_chunked" else 1024)
    a = rand_cts(o, rng, 2, L)
    da = g.to_device(a)
    out = g.alloc(2 * 2 * L * N)
    issued = g.rotate_sum(L, 2, da, steps, out)
    got = out.download((2, 2, L, N))
    unshared = 0
    for r in range(2):
        want = a[r].copy()
        for s in steps:
            w = a[r]
            terms = [s] if o.galois_elt(s) in keys else _naf_terms(s, N)
            for t in terms:
                w = o.apply_galois(w, o.galois_elt(t), keys[o.galois_elt(t)])
            unshared += len(terms) if r == 0 else 0
            want = o.add(want, w)
        assert np.array_equal(got[r], want), r
    assert issued < unshared, (issued, unshared)
    with pytest.raises(be.HE355Error):
        g.rotate_sum(L, 2, da, steps, da)  # not in place
    g.set_level_walk(True)  # the context is shared by the module's tests: back to the defaults
    g.set_latency_max(None)
    g.set_chunk(1024)


def test_partially_overlapping_outputs_are_rejected(pair, be):
    """The rotation pipelines read `in` in kernels that run after others have started writing `out` (the fused k_k3 takes polynomial 1
    of the addend from where it lies; the BFV tail reads polynomial 0 of the input through the Galois map), so ANY overlap of the two
    ranges would corrupt the input silently -- not only in == out.  apply_galois, rotate, rotate_each, rotate_sum and the BFV
    relinearize reject a shifted view of the input as output with an invalid-argument error."""
    import ctypes as C

    class View:  # a slab that starts `off` words into another one
        def __init__(self, buf, off):
            self.ptr = C.c_void_p(buf.ptr.value + off * 8)

    g, o, rng = pair
    L, N = g.L, g.N
    per = 2 * L * N
    e = o.galois_elt(1)
    g.set_galois_key(e, o.random_kswitch_key(rng))
    big = g.alloc(4 * per)
    g.fill_uniform(big, 4 * 2 * L, list(range(L)), 3)
    shifted = View(big, per)  # out = in + one ciphertext
    for call in (lambda: g.apply_galois(L, 2, big, e, shifted), lambda: g.rotate(L, 2, big, 1, shifted), lambda: g.rotate_each(L, 2, big, [1, 1], shifted),
                 lambda: g.rotate_sum(L, 2, big, [1], shifted), lambda: g.apply_galois(L, 2, shifted, e, big)):
        with pytest.raises(be.HE355Error) as ei:
            call()
        assert ei.value.code == be.E_INVALID_ARGS
    far = View(big, 2 * per)  # disjoint ranges inside one allocation are fine
    g.apply_galois(L, 2, big, e, far)
    g.sync()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 12])
def test_latency_shape_equals_throughput_shape(pair, be, n):
    """Key switches over few ciphertexts take the latency shape (he355_set_latency_max: targets of a column and digits of a tile dealt
    to more blocks, partial sums combined, unfused floor steps).  Same results as the throughput shape and as the oracle, for
    multiply -> relinearize (-> rescale), relinearize of size-3 ciphertexts, rotation and rotate_add."""
    g, o, rng = pair
    L, N = g.L, g.N
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    e1 = o.galois_elt(1)
    gk = o.random_kswitch_key(rng)
    g.set_galois_key(e1, gk)
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
    da, db = g.to_device(a), g.to_device(b)
    pw = be.Context.pairwise()
    want_rl = [o.relinearize(o.multiply_ntt(a[r], b[r]), rk) for r in range(n)]
    want_rot = [o.add(b[r], o.apply_galois(a[r], e1, gk)) for r in range(n)]
    try:
        g.set_lds_max(0)  # (rings up to N = 8192 would take the ring-in-LDS shape at these batches: test_lds_shape_* holds that one)
        for lat in (0, 16, None):  # throughput shape, latency shape, whichever the library's rule picks
            g.set_latency_max(lat)
            out = g.alloc(n * 2 * L * N)
            g.multiply_relin(L, n, da, db, pw, out)
            got = out.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], want_rl[r]), (lat, r)
            if L >= 2:
                out2 = g.alloc(n * 2 * (L - 1) * N)
                g.multiply_relin(L, n, da, db, pw, out2, rescale=True)
                got = out2.download((n, 2, L - 1, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.rescale(want_rl[r])), (lat, r)
            g.rotate_add(L, n, da, 1, db, out)
            got = out.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], want_rot[r]), (lat, r)
    finally:
        g.set_latency_max(None)
        g.set_lds_max(None)


@pytest.mark.parametrize("n", [1, 3, 20])
def test_lds_shape_equals_the_other_shapes_and_the_oracle(pair, be, n):
    """Rings that fit one CU's LDS (N <= 8192): key switches as two launches of one-polynomial workgroups (k_lds_digits, k_lds_moddown:
    he355_set_lds_max).  Same bits as the oracle and as the HBM shapes for multiply -> relinearize (-> rescale), relinearize (-> rescale) of
    size-3 ciphertexts, rotation, rotate_add (also adding in place) and a rotation through NAF steps; the path counters prove which ran."""
    g, o, rng = pair
    L, N = g.L, g.N
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    e1, e3 = o.galois_elt(1), o.galois_elt(-2)
    gk1, gk3 = o.random_kswitch_key(rng), o.random_kswitch_key(rng)
    g.set_galois_key(e1, gk1)
    g.set_galois_key(e3, gk3)
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, min(n, 2), L)
    da, db = g.to_device(a), g.to_device(b)
    ix = be.Context.outer(0, n, 0, 1)  # every a[r] times b[0]
    c3 = np.stack([o.multiply_ntt(a[r], b[0]) for r in range(n)])
    d3 = g.to_device(c3)
    want_rl = [o.relinearize(c3[r], rk) for r in range(n)]
    want_rot = [o.apply_galois(a[r], e1, gk1) for r in range(n)]
    want_rot_add = [o.add(want_rl[r], o.apply_galois(a[r], e3, gk3)) for r in range(n)]
    fits = N <= 8192 and L <= 8
    try:
        for lds in (64, 0):
            g.set_lds_max(lds)
            g.path_stats(reset=True)
            out = g.alloc(n * 2 * L * N)
            g.multiply_relin(L, n, da, db, ix, out)
            got = out.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], want_rl[r]), ("multiply_relin", lds, r)
            g.relinearize(L, n, d3, out)
            got = out.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], want_rl[r]), ("relinearize", lds, r)
            if L >= 2:
                out2 = g.alloc(n * 2 * (L - 1) * N)
                g.multiply_relin(L, n, da, db, ix, out2, rescale=True)
                got = out2.download((n, 2, L - 1, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.rescale(want_rl[r])), ("multiply_relin_rescale", lds, r)
                g.relinearize_rescale(L, n, d3, out2)
                got = out2.download((n, 2, L - 1, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.rescale(want_rl[r])), ("relinearize_rescale", lds, r)
                # he355_rescale by itself (the same floor kernel with the last data prime as its source), sizes 2 and 3
                out3 = g.alloc(n * 3 * (L - 1) * N)
                g.rescale(L, 3, n, d3, out3)
                got = out3.download((n, 3, L - 1, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.rescale(c3[r])), ("rescale size 3", lds, r)
                g.rescale(L, 2, n, da, out2)
                got = out2.download((n, 2, L - 1, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.rescale(a[r])), ("rescale size 2", lds, r)
            rot = g.alloc(n * 2 * L * N)
            g.rotate(L, n, da, 1, rot)
            got = rot.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], want_rot[r]), ("rotate", lds, r)
            # out = relinearized product; out += rotate(a, -2) IN PLACE (the addend is the output slab)
            g.relinearize(L, n, d3, out)
            g.rotate_add(L, n, da, -2, out, out)
            got = out.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], want_rot_add[r]), ("rotate_add in place", lds, r)
            st = g.path_stats()
            if lds and fits:
                assert st["ks_lds"] >= (7 if L >= 2 else 5) and st["ks_fused"] == st["ks_unfused"] == st["ks_latency"] == 0, st
            else:
                assert st["ks_lds"] == 0, st
    finally:
        g.set_lds_max(None)


@pytest.mark.parametrize("N,bits,expect_lds", [
    (4096, [60, 40, 40, 40, 40, 60], True),            # L = 5, both engines
    (2048, [46, 36, 36, 36, 36, 36, 46], True),        # L = 6: the longest chain whose partial products fit the key-switch arena
    (2048, [46, 36, 36, 36, 36, 36, 36, 46], False),   # L = 7: outside the shape -- the HBM shapes take it, same bits
])
def test_lds_shape_long_chains(be, oracle, N, bits, expect_lds):
    """The ring-in-LDS kernels at the ends of their range: every level of a 5- and a 6-prime chain (the partial-product buffer is
    (2L + 4) L N words carved from the arena behind c01), and one prime more, where the library must fall back."""
    g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    rng = np.random.default_rng(len(bits) * 1000 + N)
    try:
        g.set_lds_max(None)  # the library's own rule, whatever HE355_LDS_MAX the code-path matrix runs this module under
        rk, gk = o.random_kswitch_key(rng), o.random_kswitch_key(rng)
        e1 = o.galois_elt(1)
        g.set_relin_key(rk)
        g.set_galois_key(e1, gk)
        n = 2
        for L in range(g.L, 0, -1):
            a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
            da, db = g.to_device(a), g.to_device(b)
            g.path_stats(reset=True)
            out = g.alloc(n * 2 * L * N)
            g.multiply_relin(L, n, da, db, be.Context.pairwise(), out)
            got = out.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], o.relinearize(o.multiply_ntt(a[r], b[r]), rk)), (L, r)
            g.rotate(L, n, da, 1, out)
            got = out.download((n, 2, L, N))
            for r in range(n):
                assert np.array_equal(got[r], o.apply_galois(a[r], e1, gk)), (L, r)
            if L >= 2:
                out2 = g.alloc(n * 2 * (L - 1) * N)
                g.rescale(L, 2, n, da, out2)
                got = out2.download((n, 2, L - 1, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.rescale(a[r])), (L, r)
            st = g.path_stats()
            assert (st["ks_lds"] >= 2) == (expect_lds or L <= 6), (L, st)
    finally:
        g.close()


def test_pipeline_regression_fixture_gpu(be, oracle):
    """The HIP path reproduces tests/golden/pipeline_sha256.json (checksums of the pipeline outputs on seeded inputs, generated by
    tests/golden/make_pipeline_vectors.py): the committed fixture both the oracle (CPU suite) and the device are held to."""
    import hashlib
    import importlib.util
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_pipeline_vectors", os.path.join(here, "golden", "make_pipeline_vectors.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = json.load(open(os.path.join(here, "golden", "pipeline_sha256.json")))

    def sha(buf, shape):
        return hashlib.sha256(np.ascontiguousarray(buf.download(shape), dtype=np.uint64).tobytes()).hexdigest()

    for case in gen.CASES:
        ckks = case["scheme"] == "ckks"
        o = oracle.Context(oracle.SCHEME_CKKS if ckks else oracle.SCHEME_BFV, case["N"], bit_sizes=case["bits"], plain_bits=case.get("plain_bits", 0), sec128=False)
        g = be.Context(be.SCHEME_CKKS if ckks else be.SCHEME_BFV, case["N"], bit_sizes=case["bits"], plain_bits=case.get("plain_bits", 0), sec128=False, device=0)
        exp = want[case["name"]]
        assert [int(q) for q in g.moduli] == exp["moduli"]
        L, a, b, rk, gk = gen.inputs(o, case)
        n, N = case["n"], case["N"]
        da, db = g.to_device(a), g.to_device(b)
        g.set_relin_key(rk)
        elt = g.galois_elt(1)
        g.set_galois_key(elt, gk)
        pw = be.Context.pairwise()
        out = g.alloc(n * 2 * L * N)
        if ckks:
            g.multiply_relin(L, n, da, db, pw, out)
            assert sha(out, (n, 2, L, N)) == exp["multiply_relin"], case["name"]
            out2 = g.alloc(n * 2 * (L - 1) * N)
            g.multiply_relin(L, n, da, db, pw, out2, rescale=True)
            assert sha(out2, (n, 2, L - 1, N)) == exp["multiply_relin_rescale"], case["name"]
        else:
            c3 = g.alloc(n * 3 * L * N)
            g.bfv_multiply(L, n, da, db, pw, c3)
            g.relinearize(L, n, c3, out)
            assert sha(out, (n, 2, L, N)) == exp["bfv_multiply_relin"], case["name"]
        g.add(L, 2, n, da, db, pw, out)
        assert sha(out, (n, 2, L, N)) == exp["add"], case["name"]
        g.apply_galois(L, n, da, elt, out)
        assert sha(out, (n, 2, L, N)) == exp["rotate_1"], case["name"]
        g.close()


@pytest.mark.parametrize("name", ["ckks_n1024_60_40_60", "ckks_n1024_50_45_45_50", "ckks_n2048_60_45_45_60", "bfv_n1024_60_40_60",
                                  "bfv_n2048_60_40_40_60"])
def test_exact_model_fixture_gpu(be, name):
    """The HIP path against tests/golden/exact_vectors.json: expected outputs derived by the exact big-integer model
    (tests/golden/exact_model.py: CRT composition, exact floors, no RNS shortcuts, no oracle code).  The oracle is held to the same
    file on the CPU (tests/test_exact_model.py); here neither the oracle nor any CPU path is involved.  Step-by-step calls and the
    fused sequences (multiply_relin with and without rescale, relinearize_rescale) must all land on the model's ciphertexts."""
    import test_exact_model as tem
    f, d = tem.case_inputs(name)
    ckks = f["scheme"] == "ckks"
    N, L = f["N"], d["Ltop"]
    g = be.Context(be.SCHEME_CKKS if ckks else be.SCHEME_BFV, N, bit_sizes=f["bits"], plain_bits=0 if ckks else 20, sec128=False, device=0)
    assert [int(q) for q in g.moduli] == d["primes"]
    pw = be.Context.pairwise()
    g.set_relin_key(d["rk"])
    for elt, key in d["gk"].items():
        g.set_galois_key(elt, key)

    class HipOps:
        def add(self, a, b):
            out = g.alloc(a.size)
            g.add(a.shape[1], a.shape[0], 1, g.to_device(a[None]), g.to_device(b[None]), pw, out)
            return out.download(a.shape)

        def multiply(self, a, b):  # CKKS: dyadic tensor; BFV: BEHZ, held to the integer formula of exact_model.bfv_multiply
            out = g.alloc(3 * a.shape[1] * N)
            (g.multiply if ckks else g.bfv_multiply)(a.shape[1], 1, g.to_device(a[None]), g.to_device(b[None]), pw, out)
            return out.download((3, a.shape[1], N))

        def relinearize(self, c3, rk):
            out = g.alloc(2 * c3.shape[1] * N)
            g.relinearize(c3.shape[1], 1, g.to_device(np.ascontiguousarray(c3)[None]), out)
            return out.download((2, c3.shape[1], N))

        def rescale(self, ct):
            out = g.alloc(ct.shape[0] * (ct.shape[1] - 1) * N)
            g.rescale(ct.shape[1], ct.shape[0], 1, g.to_device(ct[None]), out)
            return out.download((ct.shape[0], ct.shape[1] - 1, N))

        def apply_galois(self, ct, elt, key):
            out = g.alloc(ct.size)
            g.apply_galois(ct.shape[1], 1, g.to_device(ct[None]), elt, out)
            return out.download(ct.shape)

    seen = set()
    for opname, got in tem.run_ops(f, d, HipOps()):
        tem.check(f, opname, got)
        seen.add(opname)
    assert seen == set(f["expected"])
    da, db = g.to_device(d["a"][None]), g.to_device(d["b"][None])
    if ckks:  # the fused kernel sequences
        out = g.alloc(2 * L * N)
        g.multiply_relin(L, 1, da, db, pw, out)
        tem.check(f, "multiply_relin", out.download((2, L, N)))
        out2 = g.alloc(2 * (L - 1) * N)
        g.multiply_relin(L, 1, da, db, pw, out2, rescale=True)
        tem.check(f, "multiply_relin_rescale", out2.download((2, L - 1, N)))
        c3 = g.alloc(3 * L * N)
        g.multiply(L, 1, da, db, pw, c3)
        g.relinearize_rescale(L, 1, c3, out2)
        tem.check(f, "multiply_relin_rescale", out2.download((2, L - 1, N)))
    if not ckks:
        # the BEHZ multiply's other launch shapes against the model: an outer product whose operands are extended and transformed once
        # each (2 x 2 results from the lists [a, a] and [b, b]: every result is a x b), and the matrix-product entry over an inner index
        # of 3 (out = 3 relin(a x b): the model's relinearized product added three times)
        a2, b2 = g.to_device(np.stack([d["a"], d["a"]])), g.to_device(np.stack([d["b"], d["b"]]))
        o4 = g.alloc(4 * 3 * L * N)
        g.bfv_multiply(L, 4, a2, b2, be.Context.outer(0, 2, 0, 2), o4)
        for r, got in enumerate(o4.download((4, 3, L, N))):
            tem.check(f, "bfv_multiply", got)
        a3, b3 = g.to_device(np.stack([d["a"]] * 3)), g.to_device(np.stack([d["b"]] * 3))
        acc = g.alloc(2 * L * N)
        g.bfv_multiply_relin_accumulate(L, 1, 1, 3, a3, 1, 1, b3, 1, 1, acc)
        c3 = g.alloc(3 * L * N)
        g.bfv_multiply(L, 1, da, db, pw, c3)
        one = g.alloc(2 * L * N)
        g.relinearize(L, 1, c3, one)
        rl = one.download((2, L, N))
        tem.check(f, "bfv_multiply_relin", rl)
        q = np.array(d["primes"][:L], dtype=object)[None, :, None]
        assert np.array_equal(acc.download((2, L, N)), ((3 * rl.astype(object)) % q).astype(np.uint64))
    # Evaluator::rotate_internal without a key for step 3: the NAF terms -1, +4 through he355_rotate
    g3 = be.Context(be.SCHEME_CKKS if ckks else be.SCHEME_BFV, N, bit_sizes=f["bits"], plain_bits=0 if ckks else 20, sec128=False, device=0)
    for elt, key in d["gk"].items():
        if elt != g3.galois_elt(3):
            g3.set_galois_key(elt, key)
    out = g3.alloc(2 * L * N)
    g3.rotate(L, 1, g3.to_device(d["a"][None]), 3, out)
    tem.check(f, "rotate_3_naf", out.download((2, L, N)))
    g3.close()
    g.close()


@pytest.mark.parametrize("name", ["ckks_n32768_60_45x15_60", "ckks_n16384_60_45x7_60"])
def test_exact_model_big_fixture_gpu(be, name):
    """The HIP path at the sizes the bench runs against tests/golden/exact_vectors_big.json (exact big-integer model, no oracle):
    N = 2^15 with the headline chain -- the fused multiply -> relinearize -> rescale sequence is the kernels of the headline
    (k_k1, k_k2n<5>, k_k3<.., fused>, k_floor_colsn<5, merged>) -- and N = 2^14, L = 8 (configs[1]'s multiply)."""
    import test_exact_model as tem
    if name not in tem.BIG:
        pytest.skip("fixture case not generated (tests/golden/make_exact_vectors_big.py)")
    f, d = tem.big_case_inputs(name)
    N, L = f["N"], d["Ltop"]
    exp = f["expected"]
    g = be.Context(be.SCHEME_CKKS, N, bit_sizes=f["bits"], device=0)
    assert [int(q) for q in g.moduli] == d["primes"]
    pw = be.Context.pairwise()
    da, db = g.to_device(d["a"][None]), g.to_device(d["b"][None])
    c3 = g.alloc(3 * L * N)
    g.multiply(L, 1, da, db, pw, c3)
    tem.check(f, "multiply", c3.download((3, L, N)))
    if "multiply_relin" in exp:
        g.set_relin_key(d["rk"])
        out = g.alloc(2 * L * N)
        g.multiply_relin(L, 1, da, db, pw, out)
        tem.check(f, "multiply_relin", out.download((2, L, N)))
        g.relinearize(L, 1, c3, out)
        tem.check(f, "multiply_relin", out.download((2, L, N)))
        out2 = g.alloc(2 * (L - 1) * N)
        g.multiply_relin(L, 1, da, db, pw, out2, rescale=True)   # batch 1: the latency shape (digit-split k_k3 + combine, unfused floor steps)
        tem.check(f, "multiply_relin_rescale", out2.download((2, L - 1, N)))
        g.set_latency_max(0)                                       # from here on the throughput shape: the headline's fused kernels
        g.multiply_relin(L, 1, da, db, pw, out)
        tem.check(f, "multiply_relin", out.download((2, L, N)))
        g.multiply_relin(L, 1, da, db, pw, out2, rescale=True)
        tem.check(f, "multiply_relin_rescale", out2.download((2, L - 1, N)))
        # the same op inside a batch that spans chunks and both streams: 5 copies of the pair, chunk 2
        g.set_chunk(2)
        a5, b5 = g.to_device(np.repeat(d["a"][None], 5, axis=0)), g.to_device(np.repeat(d["b"][None], 5, axis=0))
        out5 = g.alloc(5 * 2 * (L - 1) * N)
        g.multiply_relin(L, 5, a5, b5, pw, out5, rescale=True)
        got = out5.download((5, 2, L - 1, N))
        for r in range(5):
            tem.check(f, "multiply_relin_rescale", got[r])
    if "rotate_1" in exp:
        g.set_galois_key(d["g1"], d["gk1"])
        out = g.alloc(2 * L * N)
        for lat in (0, 4):  # throughput shape, latency shape
            g.set_latency_max(lat)
            g.apply_galois(L, 1, da, d["g1"], out)
            tem.check(f, "rotate_1", out.download((2, L, N)))
    g.close()


def test_exact_model_big_fixture_gpu_headline_launch_shape(be):
    """The headline's launch shape itself -- 1024 ciphertext pairs in ONE chunk (op-groups per block, block rounds and k_k2n target
    groups as batch 1024 selects them) -- held to the exact big-integer model: the fixture's pair replicated 1024 times on the device
    (b as the one shared operand of the outer product), every one of the 1024 results of multiply -> relinearize -> rescale must be the
    model's ciphertext.  No oracle involved."""
    import test_exact_model as tem
    name = "ckks_n32768_60_45x15_60"
    if name not in tem.BIG:
        pytest.skip("fixture case not generated (tests/golden/make_exact_vectors_big.py)")
    f, d = tem.big_case_inputs(name)
    N, L, n = f["N"], d["Ltop"], 1024
    g = be.Context(be.SCHEME_CKKS, N, bit_sizes=f["bits"], device=0)
    g.set_relin_key(d["rk"])
    one = g.to_device(d["a"][None])
    da = g.alloc(n * 2 * L * N)
    for r in range(n):
        one.copy_into(da, r * 2 * L * N)
    db = g.to_device(d["b"][None])
    out = g.alloc(n * 2 * (L - 1) * N)
    g.multiply_relin(L, n, da, db, be.Context.outer(0, n, 0, 1), out, rescale=True)
    got = out.download((n, 2, L - 1, N))
    tem.check(f, "multiply_relin_rescale", got[0])
    assert (got == got[0][None]).all(), "results of the 1024-op launch differ from each other"
    g.close()


def test_exact_model_big_fixture_gpu_bfv(be):
    """BASELINE configs[4]'s kernel instances against the exact big-integer model, no oracle involved: N = 2^15, {60, 40, 40, 60},
    t = 786433 -- he355_bfv_multiply (k_behz_extend, k_tensor4, k_behz_floor_sk), he355_relinearize (the BFV key switch and its
    coefficient-form tail), he355_apply_galois for rotate_rows(1) and rotate_columns (k_bfv_galois), and the rotation with the
    add_inplace folded in (he355_rotate_add), at batch 1 and inside a batch of 3.  Reference chain: bfv row .cpp:515-531."""
    import test_exact_model as tem
    name = "bfv_n32768_60_40_40_60"
    if name not in tem.BIG:
        pytest.skip("fixture case not generated (tests/golden/make_exact_vectors_big.py)")
    f, d = tem.big_case_inputs(name)
    N, L = f["N"], d["Ltop"]
    g = be.Context(be.SCHEME_BFV, N, bit_sizes=f["bits"], plain_bits=20, device=0)
    assert [int(q) for q in g.moduli] == d["primes"] and int(g.t) == f["plain_modulus"]
    pw = be.Context.pairwise()
    g.set_relin_key(d["rk"])
    g.set_galois_key(d["g1"], d["gk1"])
    g.set_galois_key(d["gconj"], d["gkc"])
    for n in (1, 3):
        da, db = g.to_device(np.repeat(d["a"][None], n, axis=0)), g.to_device(np.repeat(d["b"][None], n, axis=0))
        m3, rl, rot = g.alloc(n * 3 * L * N), g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
        g.bfv_multiply(L, n, da, db, pw, m3)
        got = m3.download((n, 3, L, N))
        for r in range(n):
            tem.check(f, "bfv_multiply", got[r])
        g.relinearize(L, n, m3, rl)
        got = rl.download((n, 2, L, N))
        for r in range(n):
            tem.check(f, "bfv_multiply_relin", got[r])
        g.apply_galois(L, n, rl, d["g1"], rot)
        got = rot.download((n, 2, L, N))
        for r in range(n):
            tem.check(f, "bfv_multiply_relin_rotate_rows_1", got[r])
        g.rotate(L, n, rl, 1, rot)  # the same through Evaluator::rotate_rows' step -> element mapping
        assert tem.sha(rot.download((n, 2, L, N))[0]) == f["expected"]["bfv_multiply_relin_rotate_rows_1"]["sha256"]
        g.apply_galois(L, n, da, d["gconj"], rot)
        got = rot.download((n, 2, L, N))
        for r in range(n):
            tem.check(f, "rotate_columns", got[r])
        # rotate + add_inplace as one pipeline: (rl + rotate_rows(rl, 1)) == the model's rotation plus rl, added on the host
        acc = g.alloc(n * 2 * L * N)
        g.rotate_add(L, n, rl, 1, rl, acc)
        want = np.array(f["expected"]["bfv_multiply_relin_rotate_rows_1"]["head"], dtype=np.uint64)  # heads only: the sum is checked on the first coefficients
        rlh = rl.download((n, 2, L, N))[0][:, :, :3]
        q = np.array(d["primes"][:L], dtype=np.uint64)[None, :, None]
        assert np.array_equal(acc.download((n, 2, L, N))[0][:, :, :3], (want + rlh) % q)
        # the same sum through he355_rotate_sum: node by node on the coefficient-form kernels, and level by level -- the NTT-domain walk
        # with grouped key switches that BASELINE configs[4] runs (a BFV batch transformed on the way in and out)
        g.rotate(L, n, rl, 1, rot)  # (checked against the model's digest above): the whole expected sum, every coefficient
        full_want = (rot.download((n, 2, L, N)) + rl.download((n, 2, L, N))) % q[None]
        for walk in (False, True):
            g.set_level_walk(walk)
            g.set_latency_max(0)
            g.rotate_sum(L, n, rl, [1], acc)
            assert np.array_equal(acc.download((n, 2, L, N)), full_want), walk
        g.set_level_walk(True)
        g.set_latency_max(None)
    g.close()


def test_shards_hold_the_global_batch_and_replicated_keys_agree(be):
    """Multi-GPU bench path (bench.py, sharding.shard_outer_product): a rank's shard, filled with its offset into the global operand
    array (he355_fill_uniform_at), holds exactly the rows the whole array holds there; two contexts that build their synthetic
    relinearization key from the same seed compute bit-identical results (keys are replicated by construction: the generators are
    pure functions of seed and index), and a sharded job's results are the slices of the unsharded job's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("he355_sharding_t", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                  "reference-seal-backend_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    sys.modules["he355_sharding_t"] = sharding
    spec.loader.exec_module(sharding)
    N, bits, b0 = 4096, [60, 45, 45, 60], 7
    whole = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
    L = whole.L
    pm = list(range(L))
    da, db = whole.alloc(b0 * 2 * L * N), whole.alloc(2 * L * N)
    whole.fill_uniform(da, b0 * 2 * L, pm, 1234)
    whole.fill_uniform(db, 2 * L, pm, 99)
    whole.set_relin_key_synthetic(7)
    out = whole.alloc(b0 * 2 * (L - 1) * N)
    whole.multiply_relin(L, b0, da, db, be.Context.outer(0, b0, 0, 1), out, rescale=True)
    a_all, r_all = da.download((b0, 2, L, N)), out.download((b0, 2, L - 1, N))
    for world in (2, 3):
        for rank in range(world):
            sh = sharding.shard_outer_product(b0, 1, world, rank)
            g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)  # another "device": its own context, keys, streams
            sa, sb = g.alloc(max(1, sh.a_count) * 2 * L * N), g.alloc(2 * L * N)
            g.fill_uniform(sa, sh.a_count * 2 * L, pm, 1234, first_poly=sh.a_base * 2 * L)
            g.fill_uniform(sb, 2 * L, pm, 99)
            g.set_relin_key_synthetic(7)
            assert np.array_equal(sa.download_head((sh.a_count, 2, L, N)), a_all[sh.a_base:sh.a_base + sh.a_count])
            so = g.alloc(max(1, sh.n_results) * 2 * (L - 1) * N)
            g.multiply_relin(L, sh.n_results, sa, sb, be.Context.outer(0, sh.a_count, 0, 1), so, rescale=True)
            assert np.array_equal(so.download_head((sh.n_results, 2, L - 1, N)), r_all[sh.first_result:sh.first_result + sh.n_results]), (world, rank)
            g.close()
    whole.close()


def test_chunk_shrinks_until_the_scratch_arenas_fit(be, oracle):
    """The default chunk is the whole batch up to 1024 ciphertexts (scratch 117 MiB per op at N=2^15, L=16): when the device memory that is
    free at the call cannot hold the arenas, the chunk is halved until they fit (DeviceContext::chunk_ops) and the results do not change."""
    N, bits = 32768, [60, 45, 45, 45, 60]
    g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
    o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    rng = np.random.default_rng(20261004)
    L, n = g.L, 200
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    base = rand_cts(o, rng, 4, L)
    a = np.concatenate([base[:2]] * (n // 2))  # few distinct rows, many ops
    b = np.concatenate([base[2:]] * (n // 2))
    da, db = g.to_device(a), g.to_device(b)
    out = g.alloc(n * 2 * (L - 1) * N)
    g.multiply_relin(L, n, da, db, be.Context.pairwise(), out, rescale=True)  # one chunk of 200: ~3.7 GB of scratch
    ref = out.download((n, 2, L - 1, N))
    for r in range(2):
        assert np.array_equal(ref[r], o.rescale(o.relinearize(o.multiply_ntt(a[r], b[r]), rk))), r
    g2 = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)  # a fresh context: no arena yet
    g2.set_relin_key(rk)
    da2, db2 = g2.to_device(a), g2.to_device(b)
    out2 = g2.alloc(n * 2 * (L - 1) * N)
    free, total = g2.mem_info()
    keep = 3 << 29  # leave 1.5 GiB: the arena of 200 ops (3.7 GB) does not fit, the two of 25 ops each (0.93 GB) do
    hog = g2.alloc((free - keep) // 8) if free > 2 * keep else None
    assert hog is not None
    free2, _ = g2.mem_info()
    assert free2 < 2 * keep
    g2.multiply_relin(L, n, da2, db2, be.Context.pairwise(), out2, rescale=True)
    got = out2.download((n, 2, L - 1, N))
    assert np.array_equal(got, ref)
    g2.close()
    g.close()


def test_multiply_relin_into_an_operand_slab(be, oracle):
    """he355_multiply_relin with `out` = the slab of operand 0 (pairwise, no rescale: same shape): the fused k_k3 reads the operand rows
    while it writes results, so this call takes the path on which k_k1 forms the tensor first -- and still equals the oracle."""
    g, o = make_pair(be, oracle, "n8192_default")
    rng = np.random.default_rng(77)
    L, n = g.L, 12  # throughput shape
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
    da, db = g.to_device(a), g.to_device(b)
    g.multiply_relin(L, n, da, db, be.Context.pairwise(), da)
    got = da.download((n, 2, L, g.N))
    for r in range(n):
        assert np.array_equal(got[r], o.relinearize(o.multiply_ntt(a[r], b[r]), rk)), r
    g.close()
