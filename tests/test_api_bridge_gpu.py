"""End-to-end through the HEBench API-Bridge C ABI on the MI355X: the call sequence test_harness performs
(SURVEY.md §3.5), results checked against cleartext ground truth like the harness does.  CKKS results are
approximate by construction of the scheme: tolerance 1e-4 absolute on values in [-1,1] products/sums
(45/40-bit scales give ~1e-7); BFV results are exact."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

from hebench_harness import Backend, LATENCY, OFFLINE, SCHEME_BFV, SCHEME_CKKS, W_ADD, W_DOT, W_MATMUL, W_MUL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def backend():
    be = importlib.import_module("reference-seal-backend_amd")
    if be.device_count() < 1:
        pytest.fail("no HIP device")
    b = Backend(be.LIB_PATH)
    yield b
    b.close()


def ckks_params(n, N=8192, depth=2, bits=45, scale=45):
    return [("n", n), ("PolyModulusDegree", N), ("MultiplicativeDepth", depth), ("CoefficientModulusBits", bits), ("ScaleBits", scale), ("NumThreads", 0)]


def test_ckks_eltwise_add_offline(backend):
    rng = np.random.default_rng(1234)  # CI seed of the reference (.github/workflows/cmake.yml:43)
    n = 1000
    a, b = rng.uniform(-1, 1, (3, n)), rng.uniform(-1, 1, (2, n))
    hb = backend.create(backend.find(W_ADD, SCHEME_CKKS, OFFLINE), ckks_params(n), (3, 2))
    res = backend.run(hb, [a, b], n, np.float64)
    want = (a[:, None, :] + b[None, :, :]).reshape(6, n)  # r = i*b1 + x
    assert np.allclose(res, want, atol=1e-4)
    backend.destroy(hb)


def test_ckks_eltwise_multiply_offline_size3_results(backend):
    rng = np.random.default_rng(1)
    n = 1000
    a, b = rng.uniform(-1, 1, (2, n)), rng.uniform(-1, 1, (3, n))
    hb = backend.create(backend.find(W_MUL, SCHEME_CKKS, OFFLINE), ckks_params(n), (2, 3))
    res = backend.run(hb, [a, b], n, np.float64)
    want = (a[:, None, :] * b[None, :, :]).reshape(6, n)
    assert np.allclose(res, want, atol=1e-4)
    backend.destroy(hb)


@pytest.mark.parametrize("scheme,counts", [(SCHEME_CKKS, (16, 16)), (SCHEME_CKKS, (20, 20)), (SCHEME_BFV, (20, 20))])
def test_eltwise_add_every_slot_decoded_through_the_staging_buffer(backend, scheme, counts):
    """decode() of whole-slot results: 16 x 16 results x 4096 slots x 8 B = 8 MiB fits the page-locked staging buffer createBenchmark sizes
    (256 results x n slots), 20 x 20 = 12.5 MiB does not and takes the pageable path -- same values either way (he_context.cpp: pinned)."""
    rng = np.random.default_rng(77)
    n = 4096
    if scheme == SCHEME_CKKS:
        a, b = rng.uniform(-1, 1, (counts[0], n)), rng.uniform(-1, 1, (counts[1], n))
        hb = backend.create(backend.find(W_ADD, SCHEME_CKKS, OFFLINE), ckks_params(n), counts)
        res = backend.run(hb, [a, b], n, np.float64)
        assert np.allclose(res, (a[:, None, :] + b[None, :, :]).reshape(-1, n), atol=1e-4)
    else:
        a, b = rng.integers(-1000, 1000, (counts[0], n)), rng.integers(-1000, 1000, (counts[1], n))
        hb = backend.create(backend.find(W_ADD, SCHEME_BFV, OFFLINE), bfv_params(n), counts)
        res = backend.run(hb, [a.astype(np.int64), b.astype(np.int64)], n, np.int64)
        assert np.array_equal(res, (a[:, None, :] + b[None, :, :]).reshape(-1, n))
    backend.destroy(hb)


def test_ckks_eltwise_multiply_latency_with_indexers(backend):
    rng = np.random.default_rng(2)
    n = 10
    a, b = rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (1, n))
    hb = backend.create(backend.find(W_MUL, SCHEME_CKKS, LATENCY), ckks_params(n, N=16384, depth=3))
    res = backend.run(hb, [a, b], n, np.float64)
    assert np.allclose(res[0], a[0] * b[0], atol=1e-4)
    backend.destroy(hb)


@pytest.mark.parametrize("n", [100, 128, 5])
def test_ckks_dot_product(backend, n):
    rng = np.random.default_rng(3 + n)
    a, b = rng.uniform(-1, 1, (2, n)), rng.uniform(-1, 1, (2, n))
    hb = backend.create(backend.find(W_DOT, SCHEME_CKKS, OFFLINE), ckks_params(n, bits=40, scale=40), (2, 2))
    res = backend.run(hb, [a, b], 1, np.float64)
    want = (a @ b.T).reshape(4, 1)
    assert np.allclose(res, want, atol=1e-3)
    backend.destroy(hb)


def test_bfv_eltwise_add(backend):
    rng = np.random.default_rng(4)
    n = 1000
    a, b = rng.integers(-1000, 1000, (2, n)), rng.integers(-1000, 1000, (2, n))
    hb = backend.create(backend.find(W_ADD, SCHEME_BFV, OFFLINE),
                        [("n", n), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 40), ("PlainModulusBits", 20),
                         ("NumThreads", 0)], (2, 2))
    res = backend.run(hb, [a.astype(np.int64), b.astype(np.int64)], n, np.int64)
    want = (a[:, None, :] + b[None, :, :]).reshape(4, n)
    assert np.array_equal(res, want)  # BASELINE configs[0] plumbing case (at the reference's default parameters)
    backend.destroy(hb)


def bfv_params(n, N=8192, depth=2, bits=40):
    return [("n", n), ("PolyModulusDegree", N), ("MultiplicativeDepth", depth), ("CoefficientModulusBits", bits), ("PlainModulusBits", 20), ("NumThreads", 0)]


def _centre(v, t):
    v = np.mod(v, t)
    return np.where(v > t // 2, v - t, v)


def test_bfv_eltwise_multiply(backend):
    rng = np.random.default_rng(5)
    n, t = 1000, 1032193
    a, b = rng.integers(-700, 700, (2, n)), rng.integers(-700, 700, (3, n))
    hb = backend.create(backend.find(W_MUL, SCHEME_BFV, OFFLINE), bfv_params(n), (2, 3))
    res = backend.run(hb, [a.astype(np.int64), b.astype(np.int64)], n, np.int64)
    want = _centre((a[:, None, :] * b[None, :, :]).reshape(6, n), t)
    assert np.array_equal(res, want)
    backend.destroy(hb)


@pytest.mark.parametrize("n", [100, 4096, 6000])
def test_bfv_dot_product(backend, n):
    """n = 6000 > N/2 exercises the rotate_columns branch of accumulateBFV (seal_context.cpp:305-310)."""
    rng = np.random.default_rng(6 + n)
    t = 1032193
    a, b = rng.integers(-20, 20, (2, n)), rng.integers(-20, 20, (2, n))
    hb = backend.create(backend.find(W_DOT, SCHEME_BFV, OFFLINE), bfv_params(n, bits=45), (2, 2))
    res = backend.run(hb, [a.astype(np.int64), b.astype(np.int64)], 1, np.int64)
    want = _centre((a @ b.T).reshape(4, 1), t)
    assert np.array_equal(res, want)
    backend.destroy(hb)


def _matmul(backend, dims, N, depth):
    from hebench_harness import Handle, ParameterIndexer
    import ctypes as C
    r0, c0, c1 = dims
    rng = np.random.default_rng(sum(dims))
    A, B = rng.integers(-8, 8, (r0, c0)).astype(np.int64), rng.integers(-8, 8, (c0, c1)).astype(np.int64)
    bench = [b for b in backend.benchmarks() if b["desc"].workload == W_MATMUL and b["desc"].other == 2 and b["desc"].scheme == SCHEME_BFV][0]
    hb = backend.create(bench, [("rows_M0", r0), ("cols_M0", c0), ("cols_M1", c1), ("PolyModulusDegree", N), ("MultiplicativeDepth", depth),
                                ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0)])
    res = backend.run(hb, [A.reshape(1, -1), B.reshape(1, -1)], r0 * c1, np.int64)
    backend.destroy(hb)
    t = 1032193 if N <= 8192 else 786433
    assert np.array_equal(res.reshape(r0, c1), _centre(A @ B, t))


def test_bfv_matmult_row_default_dims(backend):
    _matmul(backend, (10, 9, 8), 8192, 3)  # the reference's defaults (bfv row .cpp:41-43)


def test_bfv_matmult_row_cfg5_128(backend):
    """BASELINE configs[4]: 128x128x128 at N=2^15 (b*c = N/2 exactly): 64 row-pair ciphertexts, 127 rotations each,
    non-power-of-two steps through SEAL's NAF decomposition."""
    _matmul(backend, (128, 128, 128), 32768, 3)


def _matmul_val(backend, scheme, dims, N, depth, bits, other=0):
    r0, c0, c1 = dims
    rng = np.random.default_rng(sum(dims) + scheme)
    bench = [b for b in backend.benchmarks() if b["desc"].workload == W_MATMUL and b["desc"].other == other and b["desc"].scheme == scheme][0]
    last = ("ScaleBits", bits) if scheme == SCHEME_CKKS else ("PlainModulusBits", 20)
    hb = backend.create(bench, [("rows_M0", r0), ("cols_M0", c0), ("cols_M1", c1), ("PolyModulusDegree", N), ("MultiplicativeDepth", depth),
                                ("CoefficientModulusBits", bits), last, ("NumThreads", 0)])
    if scheme == SCHEME_CKKS:
        A, B = rng.uniform(-1, 1, (r0, c0)), rng.uniform(-1, 1, (c0, c1))
        res = backend.run(hb, [A.reshape(1, -1), B.reshape(1, -1)], r0 * c1, np.float64)
        assert np.allclose(res.reshape(r0, c1), A @ B, atol=1e-3)
    else:
        A, B = rng.integers(-8, 8, (r0, c0)).astype(np.int64), rng.integers(-8, 8, (c0, c1)).astype(np.int64)
        res = backend.run(hb, [A.reshape(1, -1), B.reshape(1, -1)], r0 * c1, np.int64)
        assert np.array_equal(res.reshape(r0, c1), _centre(A @ B, 1032193))
    backend.destroy(hb)


@pytest.mark.parametrize("dims", [(10, 9, 8), (3, 64, 5), (2, 1, 2)])
def test_ckks_matmult_val(backend, dims):
    """One row per ciphertext, (i, j) pairs as one outer-product batch (ckks matmultval .cpp:246-258); the
    reference's default dimensions and parameters first."""
    _matmul_val(backend, SCHEME_CKKS, dims, 8192, 2, 45)


@pytest.mark.parametrize("dims", [(10, 9, 8), (3, 5000, 2)])
def test_bfv_matmult_val(backend, dims):
    """cols_M0 = 5000 > N/2 takes the rotate_columns branch of accumulateBFV."""
    _matmul_val(backend, SCHEME_BFV, dims, 8192, 2, 40)


@pytest.mark.parametrize("dims", [(10, 9, 8), (2, 1, 3), (1, 17, 1)])
def test_ckks_matmult_cipher_batch_axis(backend, dims):
    """One element per ciphertext: size-3 sum over the inner dimension, then relinearize + rescale
    (ckks cipherbatchaxis .cpp:404-437), at the reference's defaults (depth 3)."""
    _matmul_val(backend, SCHEME_CKKS, dims, 8192, 3, 45, other=1)


@pytest.mark.parametrize("dims", [(10, 9, 8), (2, 1, 3)])
def test_bfv_matmult_cipher_batch_axis(backend, dims):
    _matmul_val(backend, SCHEME_BFV, dims, 8192, 3, 40, other=1)


@pytest.mark.parametrize("dims", [(10, 9, 8), (3, 64, 64), (5, 1, 7)])
def test_ckks_matmult_row(backend, dims):
    """One row per ciphertext, rotate_vector by j * (slots / cols_M0) (ckks row .cpp:472-523); 64 x 64 fills
    all 4096 slots and takes 63 rotations, most of them NAF-decomposed."""
    _matmul_val(backend, SCHEME_CKKS, dims, 8192, 3, 45, other=2)


def _sigmoid_poly(x):
    return 0.5 + 0.15012 * x - 0.0015930078125 * x ** 3  # SigmoidPolyCoeff, logreg .h:117


@pytest.mark.parametrize("category,batch,n", [(LATENCY, 1, 16), (OFFLINE, 7, 16), (OFFLINE, 20, 5)])
def test_ckks_logreg_horner(backend, category, batch, n):
    """LogisticRegression_PolyD3 at the reference's default parameters (N=16384, depth 6, 45-bit): dot products as one
    batch, collapse with rotations by -i, bias, degree-3 Horner (logreg .cpp:388-481).  Tolerance 1e-3: the scheme's
    own approximation error after five rescales with manually pinned scales (seal_context.cpp:395,452)."""
    from hebench_harness import W_LOGREG3
    rng = np.random.default_rng(100 + batch + n)
    W, b, X = rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (1, 1)), rng.uniform(-1, 1, (batch, n))
    bench = backend.find(W_LOGREG3, SCHEME_CKKS, category)
    hb = backend.create(bench, [("n", n), ("PolyModulusDegree", 16384), ("MultiplicativeDepth", 6), ("CoefficientModulusBits", 45), ("ScaleBits", 45),
                                ("NumThreads", 0)], (1, 1, batch))
    res = backend.run(hb, [W, b, X], 1, np.float64)
    want = _sigmoid_poly(X @ W[0] + b[0, 0])
    assert res.shape == (batch, 1)
    assert np.allclose(res[:, 0], want, atol=1e-3), np.abs(res[:, 0] - want).max()
    backend.destroy(hb)


@pytest.mark.parametrize("scheme", [SCHEME_CKKS, SCHEME_BFV])
def test_client_side_on_device_equals_client_side_on_host(backend, scheme):
    """encrypt()/decrypt() run on the MI355X by default and on the host with HE355_DEVICE_CLIENT=0: same keys, same
    randomness counters => the decoded results are identical to the last bit (the ciphertexts are, tests/test_gpu_client.py)."""
    rng = np.random.default_rng(77)
    n = 64
    res = []
    for flag in ("1", "0"):
        os.environ["HE355_DEVICE_CLIENT"] = flag
        try:
            if scheme == SCHEME_CKKS:
                if not res:
                    a, b = rng.uniform(-1, 1, (3, n)), rng.uniform(-1, 1, (2, n))
                hb = backend.create(backend.find(W_MUL, SCHEME_CKKS, OFFLINE), ckks_params(n), (3, 2))
                res.append(backend.run(hb, [a, b], n, np.float64))
            else:
                if not res:
                    a, b = rng.integers(-500, 500, (3, n)), rng.integers(-500, 500, (2, n))
                hb = backend.create(backend.find(W_MUL, SCHEME_BFV, OFFLINE), bfv_params(n), (3, 2))
                res.append(backend.run(hb, [a.astype(np.int64), b.astype(np.int64)], n, np.int64))
            backend.destroy(hb)
        finally:
            os.environ.pop("HE355_DEVICE_CLIENT", None)
    assert np.array_equal(res[0], res[1])
    want = (a[:, None, :] * b[None, :, :]).reshape(6, n)
    if scheme == SCHEME_CKKS:
        assert np.allclose(res[0], want, atol=1e-4)
    else:
        assert np.array_equal(res[0], _centre(want, 1032193))


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_operate_spread_over_a_device_group(backend, monkeypatch, ndev):
    """One operate() over NumDevices GPUs inside one process (csrc/bridge/multi_device.h): contiguous blocks of operand-0 rows per
    device, keys generated on every device from the shared seed, operands replicated at load(), result parts gathered at store().
    HE355_LOGICAL_DEVICES lets this one-GPU box run a 2- and a 3-device group (logical device d -> physical d mod 1): contexts, keys,
    replicas, host threads, parts and the gather are all exercised; results must equal the single-device run's and the cleartext's.
    Uneven splits (5 rows over 3 devices; 2 rows over 3 devices leaves one device idle) and all three vector workloads."""
    monkeypatch.setenv("HE355_LOGICAL_DEVICES", "3")
    rng = np.random.default_rng(40 + ndev)
    n = 64
    a, b = rng.uniform(-1, 1, (5, n)), rng.uniform(-1, 1, (2, n))
    for w, want, tol in ((W_ADD, (a[:, None, :] + b[None, :, :]).reshape(10, n), 1e-4), (W_MUL, (a[:, None, :] * b[None, :, :]).reshape(10, n), 1e-4)):
        hb = backend.create(backend.find(w, SCHEME_CKKS, OFFLINE), ckks_params(n) + [("NumDevices", ndev)], (5, 2))
        res = backend.run(hb, [a, b], n, np.float64)
        assert np.allclose(res, want, atol=tol), (w, ndev)
        backend.destroy(hb)
    hb = backend.create(backend.find(W_DOT, SCHEME_CKKS, OFFLINE), ckks_params(n, bits=40, scale=40) + [("NumDevices", ndev)], (2, 2))
    res = backend.run(hb, [a[:2], b], 1, np.float64)
    assert np.allclose(res, (a[:2] @ b.T).reshape(4, 1), atol=1e-3), ndev
    backend.destroy(hb)
    # BFV: exact integers, so the group's result must equal the cleartext bit for bit
    x, y = rng.integers(-500, 500, (4, n)), rng.integers(-500, 500, (3, n))
    bfv = [("n", n), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0),
           ("NumDevices", ndev)]
    hb = backend.create(backend.find(W_MUL, SCHEME_BFV, OFFLINE), bfv, (4, 3))
    res = backend.run(hb, [x, y], n, np.int64)
    assert np.array_equal(res, (x[:, None, :] * y[None, :, :]).reshape(12, n)), ndev
    backend.destroy(hb)
    txt = backend.description_text(backend.find(W_ADD, SCHEME_CKKS, OFFLINE), ckks_params(n) + [("NumDevices", ndev)])
    assert txt.rstrip().endswith(f", Number of devices, {ndev}")
    if ndev > 1:
        # load() sent every other device only ITS block of operand-0 rows (shard operand 0, broadcast operand 1): the
        # last load was the BFV one above -- 4 rows of operand 0 over ndev devices, 3 rows of operand 1 whole
        moved = backend.L.he355_bridge_group_load_bytes
        moved.restype, moved.argtypes = C.c_uint64, [C.c_int, C.c_int]
        ct = 2 * 2 * 8192 * 8  # size-2 ciphertext at L = 2 data primes, N = 8192
        base, extra = divmod(4, ndev)
        for d in range(1, ndev):
            assert moved(d, 0) == (base + (1 if d < extra else 0)) * ct, (d, moved(d, 0))
            assert moved(d, 1) == 3 * ct
        # an operate() that indexes a portion of the batch runs on the primary device (which holds everything) and is still right
        hb = backend.create(backend.find(W_ADD, SCHEME_CKKS, OFFLINE), ckks_params(n) + [("NumDevices", ndev)], (5, 2))
        res = backend.run(hb, [a, b], n, np.float64, indexers=[(1, 3), (0, 2)])
        assert np.allclose(res, (a[1:4, None, :] + b[None, :, :]).reshape(6, n), atol=1e-4)
        backend.destroy(hb)


def test_two_physical_devices(backend, monkeypatch):
    """The first box with two GPUs that runs `pytest -m gpu` covers the cross-device paths: he355_copy_peer in both directions between two
    PHYSICAL devices (a round trip of random bytes), and one operate() of every vector workload spread over a two-device group without
    HE355_LOGICAL_DEVICES -- contexts on devices 0 and 1, keys generated per device from the shared seed, hipMemcpyPeer at load(),
    parts gathered at store() (csrc/bridge/multi_device.cpp) -- equal to the single-device run and the cleartext.  Skipped on a
    one-GPU box (where the logical-device tests above run the same code on one card)."""
    be = importlib.import_module("reference-seal-backend_amd")
    if be.device_count() < 2:
        pytest.skip("needs two physical GPUs")
    monkeypatch.delenv("HE355_LOGICAL_DEVICES", raising=False)
    # he355_copy_peer, both directions
    g0 = be.Context(be.SCHEME_CKKS, 4096, bit_sizes=[60, 45, 60], sec128=False, device=0)
    g1 = be.Context(be.SCHEME_CKKS, 4096, bit_sizes=[60, 45, 60], sec128=False, device=1)
    rng = np.random.default_rng(77)
    x = rng.integers(0, 2 ** 63, 1 << 16, dtype=np.uint64)
    d0, d1, back = g0.to_device(x), g1.alloc(x.size), g0.alloc(x.size)
    lib = be.lib()
    assert lib.he355_copy_peer(g1.h, d1.ptr, g0.h, d0.ptr, x.nbytes) == 0
    g1.sync()
    assert np.array_equal(d1.download(x.shape), x)
    assert lib.he355_copy_peer(g0.h, back.ptr, g1.h, d1.ptr, x.nbytes) == 0
    g0.sync()
    assert np.array_equal(back.download(x.shape), x)
    g0.close(); g1.close()
    # one operate() over two physical devices, against one device and the cleartext
    n = 64
    a, b = rng.uniform(-1, 1, (5, n)), rng.uniform(-1, 1, (2, n))
    for w, want, tol in ((W_ADD, (a[:, None, :] + b[None, :, :]).reshape(10, n), 1e-4), (W_MUL, (a[:, None, :] * b[None, :, :]).reshape(10, n), 1e-4)):
        got = []
        for ndev in (1, 2):
            hb = backend.create(backend.find(w, SCHEME_CKKS, OFFLINE), ckks_params(n) + [("NumDevices", ndev)], (5, 2))
            got.append(backend.run(hb, [a, b], n, np.float64))
            backend.destroy(hb)
            assert np.allclose(got[-1], want, atol=tol), (w, ndev)
    hb = backend.create(backend.find(W_DOT, SCHEME_CKKS, OFFLINE), ckks_params(n, bits=40, scale=40) + [("NumDevices", 2)], (2, 2))
    res = backend.run(hb, [a[:2], b], 1, np.float64)
    assert np.allclose(res, (a[:2] @ b.T).reshape(4, 1), atol=1e-3)
    backend.destroy(hb)
    x, y = rng.integers(-500, 500, (4, n)), rng.integers(-500, 500, (3, n))
    bfv = [("n", n), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0),
           ("NumDevices", 2)]
    hb = backend.create(backend.find(W_MUL, SCHEME_BFV, OFFLINE), bfv, (4, 3))
    res = backend.run(hb, [x, y], n, np.int64)
    assert np.array_equal(res, (x[:, None, :] * y[None, :, :]).reshape(12, n))
    backend.destroy(hb)
    moved = backend.L.he355_bridge_group_load_bytes
    moved.restype, moved.argtypes = C.c_uint64, [C.c_int, C.c_int]
    ct = 2 * 2 * 8192 * 8
    assert moved(1, 0) == 2 * ct and moved(1, 1) == 3 * ct  # device 1 received its block of operand-0 rows and all of operand 1


@pytest.mark.parametrize("ndev", [2, 3])
def test_matmult_row_spread_over_a_device_group(backend, monkeypatch, ndev):
    """MatMultRow's row(-pair) ciphertexts are independent (bfv row .cpp:512-533): HE355_NUM_DEVICES spreads them over a device
    group in contiguous blocks, B and the keys on every device (last row).  5 rows -> 3 row-pair ciphertexts (BFV) /
    5 ciphertexts (CKKS) over 2 and 3 logical devices on this one GPU; results equal the cleartext product."""
    monkeypatch.setenv("HE355_LOGICAL_DEVICES", "3")
    monkeypatch.setenv("HE355_NUM_DEVICES", str(ndev))
    _matmul(backend, (5, 8, 4), 8192, 3)
    _matmul_val(backend, SCHEME_CKKS, (5, 8, 4), 8192, 3, 40, other=2)  # CKKS MatMultRow (one row per ciphertext)


def _steady_state_case(backend, which):
    rng = np.random.default_rng(9)
    if which == "ckks_mul_latency":
        n = 16
        return backend.create(backend.find(W_MUL, SCHEME_CKKS, LATENCY), ckks_params(n, N=16384, depth=3)), [rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (1, n))]
    if which == "ckks_dot_offline":
        n = 100
        return backend.create(backend.find(W_DOT, SCHEME_CKKS, OFFLINE), ckks_params(n, bits=40, scale=40), (3, 2)), [rng.uniform(-1, 1, (3, n)), rng.uniform(-1, 1, (2, n))]
    if which == "bfv_dot_offline":
        n = 6000  # rotate_columns branch too
        ops = [rng.integers(-20, 20, (2, n)).astype(np.int64), rng.integers(-20, 20, (2, n)).astype(np.int64)]
        return backend.create(backend.find(W_DOT, SCHEME_BFV, OFFLINE), bfv_params(n, bits=45), (2, 2)), ops
    if which == "bfv_matmult_row":
        bench = [b for b in backend.benchmarks() if b["desc"].workload == W_MATMUL and b["desc"].other == 2 and b["desc"].scheme == SCHEME_BFV][0]
        hb = backend.create(bench, [("rows_M0", 5), ("cols_M0", 8), ("cols_M1", 4), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3),
                                    ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0)])
        return hb, [rng.integers(-8, 8, (1, 40)).astype(np.int64), rng.integers(-8, 8, (1, 32)).astype(np.int64)]
    if which == "ckks_cipher_batch_axis":
        bench = [b for b in backend.benchmarks() if b["desc"].workload == W_MATMUL and b["desc"].other == 1 and b["desc"].scheme == SCHEME_CKKS][0]
        hb = backend.create(bench, [("rows_M0", 3), ("cols_M0", 4), ("cols_M1", 2), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3),
                                    ("CoefficientModulusBits", 45), ("ScaleBits", 45), ("NumThreads", 0)])
        return hb, [rng.uniform(-1, 1, (1, 12)), rng.uniform(-1, 1, (1, 8))]
    from hebench_harness import W_LOGREG3
    n, batch = 16, 5
    hb = backend.create(backend.find(W_LOGREG3, SCHEME_CKKS, OFFLINE), [("n", n), ("PolyModulusDegree", 16384), ("MultiplicativeDepth", 6),
                                                                          ("CoefficientModulusBits", 45), ("ScaleBits", 45), ("NumThreads", 0)], (1, 1, batch))
    return hb, [rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (1, 1)), rng.uniform(-1, 1, (batch, n))]


@pytest.mark.parametrize("which", ["ckks_mul_latency", "ckks_dot_offline", "bfv_dot_offline", "bfv_matmult_row", "ckks_cipher_batch_axis", "ckks_logreg_offline"])
def test_steady_state_operate_does_not_allocate(backend, which):
    """The timed region of a harness run is operate() (ckks eltwise .cpp:306-366).  Result slabs and temporaries come from the context's
    device pool (csrc/device_pool.h): after the first call has sized the arenas, a second and a third operate() -- each result handle
    destroyed in between, as the harness's measurement loop does -- perform no hipMalloc and no hipFree at all (he355_alloc_stats,
    process totals), and every he355_malloc inside them is served from a free list."""
    from hebench_harness import Handle, ParameterIndexer
    if os.environ.get("HE355_POOL", "1")[:1] == "0":
        pytest.skip("HE355_POOL=0 selects the pre-pool allocation behaviour on purpose")
    be = importlib.import_module("reference-seal-backend_amd")
    hb, operands = _steady_state_case(backend, which)
    L = backend.L
    dpc, keep = backend.pack(operands)
    h_plain, h_cipher, h_remote = Handle(), Handle(), Handle()
    backend.chk(L.encode(hb, C.byref(dpc), C.byref(h_plain)))
    backend.chk(L.encrypt(hb, h_plain, C.byref(h_cipher)))
    backend.chk(L.load(hb, C.byref(h_cipher), 1, C.byref(h_remote)))
    idx = [(0, o.shape[0]) for o in operands]
    pi = (ParameterIndexer * len(idx))(*[ParameterIndexer(v, b) for v, b in idx])
    stats = []
    for _ in range(3):
        h_out = Handle()
        backend.chk(L.operate(hb, h_remote, pi, len(idx), C.byref(h_out)))
        L.destroyHandle(h_out)
        stats.append(be.process_alloc_stats())
    assert stats[1]["raw_mallocs"] == stats[0]["raw_mallocs"] == stats[2]["raw_mallocs"], stats
    assert stats[1]["raw_frees"] == stats[0]["raw_frees"] == stats[2]["raw_frees"], stats
    assert stats[2]["pool_misses"] == stats[0]["pool_misses"], stats       # nothing new was taken from the device ...
    assert stats[2]["pool_hits"] > stats[1]["pool_hits"] > stats[0]["pool_hits"], stats  # ... and the result slabs came from the lists
    for h in (h_plain, h_cipher, h_remote):
        L.destroyHandle(h)
    backend.destroy(hb)


def test_pool_reuses_and_trims():
    """he355_malloc / he355_free: a freed block of the same size class is handed out again without a HIP call; he355_pool_trim gives the
    cached blocks back to the device; a block keeps its contents' independence (two live blocks never alias)."""
    if os.environ.get("HE355_POOL", "1")[:1] == "0":
        pytest.skip("HE355_POOL=0 selects the pre-pool allocation behaviour on purpose")
    be = importlib.import_module("reference-seal-backend_amd")
    g = be.Context(be.SCHEME_CKKS, 4096, bit_sizes=[60, 45, 60], sec128=False, device=0)
    a = g.alloc(1 << 16)
    b = g.alloc(1 << 16)
    assert a.ptr.value != b.ptr.value
    s0 = g.alloc_stats()
    pa = a.ptr.value
    a.free()
    c = g.alloc(1 << 16)
    s1 = g.alloc_stats()
    assert c.ptr.value == pa and s1["raw_mallocs"] == s0["raw_mallocs"] and s1["pool_hits"] == s0["pool_hits"] + 1
    x = np.arange(1 << 16, dtype=np.uint64)
    c.upload(x)
    b.upload(x[::-1].copy())
    assert np.array_equal(c.download(), x) and np.array_equal(b.download(), x[::-1])
    c.free()
    assert g.alloc_stats()["cached_bytes"] >= (1 << 19)
    assert g.pool_trim() >= (1 << 19)
    s2 = g.alloc_stats()
    assert s2["cached_bytes"] == 0 and s2["raw_frees"] == s1["raw_frees"] + 1
    g.close()


def test_double_free_and_foreign_pointers_are_rejected():
    """he355_free of a block that already sits on the pool's free list, or that another context allocated, is an argument error -- it
    must not be hipFree'd while the list still holds it (the next allocation of that class would receive freed memory)."""
    if os.environ.get("HE355_POOL", "1")[:1] == "0":
        pytest.skip("HE355_POOL=0 selects the pre-pool allocation behaviour on purpose")
    be = importlib.import_module("reference-seal-backend_amd")
    g = be.Context(be.SCHEME_CKKS, 4096, bit_sizes=[60, 45, 60], sec128=False, device=0)
    g2 = be.Context(be.SCHEME_CKKS, 4096, bit_sizes=[60, 45, 60], sec128=False, device=0)
    lib = be.lib()
    a = g.alloc(1 << 14)
    p = a.ptr
    assert lib.he355_free(g.h, p) == 0
    a.ptr = None
    s0 = g.alloc_stats()
    assert lib.he355_free(g.h, p) == be.E_INVALID_ARGS           # freed twice
    b = g2.alloc(1 << 14)
    assert lib.he355_free(g.h, b.ptr) == be.E_INVALID_ARGS       # another context's block
    s1 = g.alloc_stats()
    assert s1["raw_frees"] == s0["raw_frees"] and s1["cached_bytes"] == s0["cached_bytes"]
    c = g.alloc(1 << 14)                                               # the cached block is still good
    assert c.ptr.value == p.value
    x = np.arange(1 << 14, dtype=np.uint64)
    c.upload(x)
    assert np.array_equal(c.download(), x)
    g.close(); g2.close()
