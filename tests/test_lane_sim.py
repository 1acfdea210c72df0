"""CPU check of the product's lane programs (ntt_core.h) through the test-only lane simulator:
index maps, LDS exchange layouts, twiddle addressing, and the fp64 engine's magnitude bounds,
compared bit-for-bit with the oracle.  No GPU needed."""
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
    u64p = C.POINTER(C.c_uint64)
    L.sim_params_create.restype = C.c_void_p
    L.sim_params_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_int), C.c_size_t, C.c_int, C.c_int]
    L.sim_params_destroy.argtypes = [C.c_void_p]
    L.sim_modulus.restype = C.c_uint64
    L.sim_modulus.argtypes = [C.c_void_p, C.c_size_t]
    L.sim_root.restype = C.c_uint64
    L.sim_root.argtypes = [C.c_void_p, C.c_size_t]
    L.sim_is_f64.argtypes = [C.c_void_p, C.c_size_t]
    L.sim_K.restype = C.c_size_t
    L.sim_K.argtypes = [C.c_void_p]
    L.sim_maxmag_reset.restype = C.c_double
    L.sim_ntt_forward.argtypes = [C.c_void_p, C.c_size_t, u64p]
    L.sim_ntt_inverse.argtypes = [C.c_void_p, C.c_size_t, u64p]
    L.sim_galois_elt.restype = C.c_uint32
    L.sim_galois_elt.argtypes = [C.c_void_p, C.c_int]
    L.sim_galois_elts_all.restype = C.c_size_t
    L.sim_galois_elts_all.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.sim_galois_perm.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    return L


def _mk(sim, N, bits, force_u64=False):
    if force_u64:
        os.environ["HE355_FORCE_U64"] = "1"
    try:
        arr = (C.c_int * len(bits))(*bits)
        h = sim.sim_params_create(2, N, arr, len(bits), 0, 0)
    finally:
        os.environ.pop("HE355_FORCE_U64", None)
    assert h
    return h


@pytest.mark.parametrize("N", [1024, 2048, 4096, 8192, 16384, 32768])
@pytest.mark.parametrize("force_u64", [False, True])
def test_lane_program_matches_oracle(sim, oracle, N, force_u64):
    bits = [60, 45, 40, 46, 60]
    h = _mk(sim, N, bits, force_u64)
    ctx = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    rng = np.random.default_rng(N + force_u64)
    try:
        for i, q in enumerate(ctx.moduli):
            assert sim.sim_modulus(h, i) == q and sim.sim_root(h, i) == ctx.root(i)
            assert bool(sim.sim_is_f64(h, i)) == ((q < 2 ** 47) and not force_u64)
            for trial in range(2):
                a = rng.integers(0, q, N, dtype=np.uint64)
                if trial == 1:  # adversarial magnitudes: everything at q-1
                    a[:] = q - 1
                f = a.copy()
                sim.sim_ntt_forward(h, i, oracle._p(f))
                assert np.array_equal(f, ctx.ntt(i, a)), (N, i, "forward")
                g = f.copy()
                sim.sim_ntt_inverse(h, i, oracle._p(g))
                assert np.array_equal(g, a), (N, i, "inverse")
        mag = sim.sim_maxmag_reset()
        if not force_u64:
            assert 0 < mag < 2.0 ** 52, mag  # exactness bound of the fp64 engine (integers < 2^53)
    finally:
        sim.sim_params_destroy(h)


def test_galois_rules_match_oracle(sim, oracle):
    N = 4096
    bits = [50, 40, 50]
    h = _mk(sim, N, bits)
    ctx = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    try:
        buf = (C.c_uint32 * 64)()
        n = sim.sim_galois_elts_all(h, buf)
        assert list(buf[:n]) == ctx.galois_elts_all()
        for step in (0, 1, -1, 5, -7, 1024, 2047, 2048):
            assert sim.sim_galois_elt(h, step) == ctx.galois_elt(step)
        rng = np.random.default_rng(9)
        a = rng.integers(0, ctx.moduli[0], N, dtype=np.uint64)
        perm = (C.c_uint32 * N)()
        for elt in (3, 9, 2 * N - 1, ctx.galois_elt(-3)):
            sim.sim_galois_perm(h, elt, perm)
            got = a[np.frombuffer(perm, dtype=np.uint32)]
            assert np.array_equal(got, ctx.apply_galois_poly(0, elt, True, a))
    finally:
        sim.sim_params_destroy(h)
