"""Client-side host code of the product (keys, encoders, encrypt/decrypt) cross-checked with the oracle on the CPU:
keys and ciphertexts are SEAL-layout arrays, so each side must be able to consume the other's."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "csim"), "-s"], check=True)
    L = C.CDLL(os.path.join(HERE, "csim", "_build", "libcsim.so"))
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.sim_params_create.restype = vp
    L.sim_params_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_int), C.c_size_t, C.c_int, C.c_int]
    L.sim_params_destroy.argtypes = [vp]
    L.simc_create.restype = vp
    L.simc_create.argtypes = [vp, C.c_uint64]
    for name, args in {"simc_destroy": [vp], "simc_secret_key": [vp, u64p], "simc_public_key": [vp, u64p], "simc_relin_key": [vp, u64p],
                       "simc_galois_key": [vp, C.c_uint32, u64p], "simc_ckks_encode": [vp, C.POINTER(C.c_double), C.c_size_t, C.c_double, u64p],
                       "simc_ckks_decode": [vp, u64p, C.c_size_t, C.c_double, C.POINTER(C.c_double)],
                       "simc_bfv_encode": [vp, C.POINTER(C.c_int64), C.c_size_t, u64p], "simc_bfv_decode": [vp, u64p, C.POINTER(C.c_int64)],
                       "simc_encrypt": [vp, u64p, u64p], "simc_decrypt": [vp, u64p, C.c_size_t, C.c_size_t, u64p],
                       "simc_set_encrypt_index": [vp, C.c_uint64],
                       "simc_sample": [C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, C.POINTER(C.c_int32)]}.items():
        getattr(L, name).argtypes = args
    L.simc_encrypt_seed.restype = C.c_uint64
    L.simc_encrypt_seed.argtypes = [vp]
    L.simc_encrypt_index.restype = C.c_uint64
    L.simc_encrypt_index.argtypes = [vp]
    return L


def _mk(sim, scheme, N, bits, plain_bits=0):
    arr = (C.c_int * len(bits))(*bits)
    p = sim.sim_params_create(scheme, N, arr, len(bits), plain_bits, 0)
    assert p
    return p, sim.simc_create(p, 42)


def test_ckks_client_interoperates_with_oracle(sim, oracle):
    N, bits, scale = 2048, [60, 40, 40, 60], 2.0 ** 40
    p, c = _mk(sim, 2, N, bits)
    o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    K, L = o.K, o.L
    sk = np.empty((K, N), dtype=np.uint64)
    sim.simc_secret_key(c, oracle._p(sk))
    rng = np.random.default_rng(0)
    x, y = rng.uniform(-1, 1, N // 2), rng.uniform(-1, 1, N // 2)
    px, py = np.empty((L, N), dtype=np.uint64), np.empty((L, N), dtype=np.uint64)
    dp = C.POINTER(C.c_double)
    sim.simc_ckks_encode(c, x.ctypes.data_as(dp), len(x), scale, oracle._p(px))
    sim.simc_ckks_encode(c, y.ctypes.data_as(dp), len(y), scale, oracle._p(py))
    # encoders agree with the numpy restatement (same slot -> evaluation-point map)
    assert np.allclose(oracle.ckks_decode(o, px, scale).real, x, atol=1e-6)
    cx, cy = np.empty((2, L, N), dtype=np.uint64), np.empty((2, L, N), dtype=np.uint64)
    sim.simc_encrypt(c, oracle._p(px), oracle._p(cx))
    sim.simc_encrypt(c, oracle._p(py), oracle._p(cy))
    # product ciphertext + product secret key -> oracle decryption
    assert np.allclose(oracle.ckks_decode(o, o.decrypt_phase(cx, sk), scale).real, x, atol=1e-5)
    # product relin / Galois keys drive the ORACLE evaluator; the product decrypts (size 3 included)
    rk = np.empty((L, 2, K, N), dtype=np.uint64)
    sim.simc_relin_key(c, oracle._p(rk))
    c3 = o.multiply_ntt(cx, cy)
    out = np.empty(N // 2)
    dec3 = np.empty((L, N), dtype=np.uint64)
    sim.simc_decrypt(c, oracle._p(c3), 3, L, oracle._p(dec3))
    sim.simc_ckks_decode(c, oracle._p(dec3), L, scale * scale, out.ctypes.data_as(dp))
    assert np.allclose(out, x * y, atol=1e-5)
    c2 = o.rescale(o.relinearize(c3, rk))
    dec = np.empty((L - 1, N), dtype=np.uint64)
    sim.simc_decrypt(c, oracle._p(c2), 2, L - 1, oracle._p(dec))
    sim.simc_ckks_decode(c, oracle._p(dec), L - 1, scale * scale / o.moduli[L - 1], out.ctypes.data_as(dp))
    assert np.allclose(out, x * y, atol=1e-5)
    elt = o.galois_elt(3)
    gk = np.empty((L, 2, K, N), dtype=np.uint64)
    sim.simc_galois_key(c, elt, oracle._p(gk))
    r = o.apply_galois(cx, elt, gk)
    dec = np.empty((L, N), dtype=np.uint64)
    sim.simc_decrypt(c, oracle._p(r), 2, L, oracle._p(dec))
    sim.simc_ckks_decode(c, oracle._p(dec), L, scale, out.ctypes.data_as(dp))
    assert np.allclose(out, np.roll(x, -3), atol=1e-5)
    sim.simc_destroy(c)
    sim.sim_params_destroy(p)


def test_bfv_client_interoperates_with_oracle(sim, oracle):
    N, bits = 2048, [50, 40, 50]
    p, c = _mk(sim, 1, N, bits, 20)
    o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=20, sec128=False)
    codec = oracle.BatchCodec(N, o.t)
    L = o.L
    rng = np.random.default_rng(1)
    x, y = rng.integers(-500, 500, N), rng.integers(-500, 500, N)
    px, py = np.empty(N, dtype=np.uint64), np.empty(N, dtype=np.uint64)
    ip = C.POINTER(C.c_int64)
    sim.simc_bfv_encode(c, x.astype(np.int64).ctypes.data_as(ip), N, oracle._p(px))
    sim.simc_bfv_encode(c, y.astype(np.int64).ctypes.data_as(ip), N, oracle._p(py))
    assert np.array_equal(px, codec.encode(x))  # BatchEncoder restatements agree exactly
    cx, cy = np.empty((2, L, N), dtype=np.uint64), np.empty((2, L, N), dtype=np.uint64)
    sim.simc_encrypt(c, oracle._p(px), oracle._p(cx))
    sim.simc_encrypt(c, oracle._p(py), oracle._p(cy))
    s = o.add(cx, cy)
    dec = np.empty(N, dtype=np.uint64)
    sim.simc_decrypt(c, oracle._p(s), 2, L, oracle._p(dec))
    out = np.empty(N, dtype=np.int64)
    sim.simc_bfv_decode(c, oracle._p(dec), out.ctypes.data_as(ip))
    assert np.array_equal(out, x + y)
    # oracle BEHZ product of product-side ciphertexts decrypts (size 3) on the product side
    c3 = o.bfv_multiply(cx, cy)
    sim.simc_decrypt(c, oracle._p(c3), 3, L, oracle._p(dec))
    sim.simc_bfv_decode(c, oracle._p(dec), out.ctypes.data_as(ip))
    t = o.t
    want = (x * y) % t
    want = np.where(want > t // 2, want - t, want)
    assert np.array_equal(out, want)
    sim.simc_destroy(c)
    sim.sim_params_destroy(p)


def test_sampler_mirror_matches_shared_header(sim):
    """tests/sampler_np.py (numpy) == csrc/client/sampler.h (the code the host client and the device kernels run)."""
    import sampler_np as sn
    for seed, stream in ((1, 0), (0xDEADBEEFCAFEF00D, 17), (2 ** 64 - 1, 2 ** 40 + 5)):
        for kind, fn in ((0, sn.sample_ternary), (1, sn.sample_cbd)):
            got = np.empty(4096, dtype=np.int32)
            sim.simc_sample(seed, stream, 4096, kind, got.ctypes.data_as(C.POINTER(C.c_int32)))
            assert np.array_equal(got, fn(seed, stream, 4096)), (seed, stream, kind)
    t = sn.sample_ternary(5, 9, 1 << 16)
    assert set(np.unique(t)) == {-1, 0, 1} and abs(t.mean()) < 0.02
    e = sn.sample_cbd(5, 9, 1 << 16)
    assert abs(e.mean()) < 0.05 and 3.0 < e.std() < 3.5  # sigma = sqrt(21/2) = 3.24


@pytest.mark.parametrize("scheme", ["ckks", "bfv"])
def test_client_encryption_is_bit_exact_with_oracle(sim, oracle, scheme):
    """Same public key, same plaintext, same sampled polynomials => the product's host encryption equals the oracle's
    (ho_encrypt_explicit) bit for bit, including the divide-and-round by the special prime and BFV's scaling variant."""
    import sampler_np as sn
    N = 2048
    if scheme == "ckks":
        bits, pb = [60, 40, 40, 60], 0
        p, c = _mk(sim, 2, N, bits)
        o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    else:
        bits, pb = [50, 40, 50], 20
        p, c = _mk(sim, 1, N, bits, pb)
        o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False)
    K, L = o.K, o.L
    pk = np.empty((2, K, N), dtype=np.uint64)
    sim.simc_public_key(c, oracle._p(pk))
    rng = np.random.default_rng(3)
    seed = sim.simc_encrypt_seed(c)
    sim.simc_set_encrypt_index(c, 7)
    for r in range(7, 10):
        plain = o.random_poly(rng, L, 1)[0] if scheme == "ckks" else rng.integers(0, o.t, N).astype(np.uint64)
        got = np.empty((2, L, N), dtype=np.uint64)
        assert sim.simc_encrypt_index(c) == r
        sim.simc_encrypt(c, oracle._p(np.ascontiguousarray(plain)), oracle._p(got))
        su, s0, s1 = sn.enc_streams(r)
        want = o.encrypt_explicit(pk, plain, sn.sample_ternary(seed, su, N), sn.sample_cbd(seed, s0, N), sn.sample_cbd(seed, s1, N))
        assert np.array_equal(got, want), r
    sim.simc_destroy(c)
    sim.sim_params_destroy(p)
