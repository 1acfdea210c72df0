"""End-to-end through the HEBench API-Bridge C ABI on the MI355X: the call sequence test_harness performs
(SURVEY.md §3.5), results checked against cleartext ground truth like the harness does.  CKKS results are
approximate by construction of the scheme: tolerance 1e-4 absolute on values in [-1,1] products/sums
(45/40-bit scales give ~1e-7); BFV results are exact."""
import importlib
import os

import numpy as np
import pytest

from hebench_harness import Backend, LATENCY, OFFLINE, SCHEME_BFV, SCHEME_CKKS, W_ADD, W_DOT, W_MUL

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
