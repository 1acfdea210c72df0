"""CPU check of the product's lane programs (ntt_core.h) through the test-only lane simulator:
index maps, LDS exchange layouts, twiddle addressing, and the fp64 engine's magnitude bounds,
compared bit-for-bit with the oracle.  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


# both builds of the u64 engine (csrc/modarith.h): Shoup quotients, and the fold reduction for primes 2^60 - c
@pytest.fixture(scope="module", params=["shoup", "fold"])
def sim(request):
    subprocess.run(["make", "-C", os.path.join(HERE, "csim"), "-s"], check=True)
    L = C.CDLL(os.path.join(HERE, "csim", "_build", "libcsim.so" if request.param == "shoup" else "libcsim_fold.so"))
    L.sim_u64_fold_build.restype = C.c_int
    assert L.sim_u64_fold_build() == (request.param == "fold")
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
    if not h and sim.sim_u64_fold_build():
        pytest.skip("this parameter set puts a prime that is not 2^60 - c on the u64 engine: no fold form")
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
                sim.sim_overflow_reset()
                sim.sim_ntt_inverse(h, i, oracle._p(g))
                assert np.array_equal(g, a), (N, i, "inverse")
                assert sim.sim_overflow_reset() == 0, (N, i, "a value of the inverse transform left its bound")
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


@pytest.mark.parametrize("N", [1024, 2048, 8192, 32768])
def test_every_default_rotation_maps_a_row_onto_one_source_row(sim, N):
    """k_k1 (K1_GALOIS) reads the source row of a rotated row WHOLE and permutes it in LDS, and k_k3 gathers the permuted c0 from one
    8 KiB row: both rest on the NTT-domain permutation of every Galois element mapping each row of 1024 slots onto one row of the source.
    Params::galois_perm_ntt throws otherwise; here for every element of the default key set and the conjugation, and the map is checked
    to be a bijection of the rows."""
    bits = [50, 40, 50]
    h = _mk(sim, N, bits)
    try:
        buf = (C.c_uint32 * 64)()
        n = sim.sim_galois_elts_all(h, buf)
        perm = (C.c_uint32 * N)()
        for elt in list(buf[:n]) + [2 * N - 1, 3, 5 ** 3 % (2 * N)]:
            sim.sim_galois_perm(h, elt, perm)  # (throws -> a null table / abort would fail the test)
            p = np.frombuffer(perm, dtype=np.uint32).reshape(N // 1024, 1024)
            rows = p >> 10
            assert (rows == rows[:, :1]).all(), elt
            assert sorted(rows[:, 0].tolist()) == list(range(N // 1024)), elt
            assert all(sorted((r & 1023).tolist()) == list(range(1024)) for r in p), elt
    finally:
        sim.sim_params_destroy(h)


def _ntt_primes(N, lo, hi, count):
    """`count` primes 1 (mod 2N) in [lo, hi), largest first"""
    from sympy import isprime
    out, v = [], (hi - 2) // (2 * N) * (2 * N) + 1
    while v >= lo and len(out) < count:
        if isprime(v):
            out.append(v)
        v -= 2 * N
    return out


@pytest.mark.parametrize("N", [1024, 4096, 32768])
def test_fold_form_at_the_ends_of_its_range(sim, oracle, N):
    """Fold build of the u64 engine (q = 2^60 - c): primes with the largest c the form admits (just below 2^26) and the smallest, rows
    whose every element sits at the top of the lazy input range (4q - 1) and at q - 1: no sum leaves 64 bits, every phase stays
    inside its bound, and the wide-lazy row pass agrees with the Harvey row pass (the simulator throws otherwise); transforms of
    random data equal the oracle's.  The Shoup build runs the same primes through its own bounds."""
    sim.sim_params_create_primes.restype = C.c_void_p
    sim.sim_params_create_primes.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64]
    sim.sim_row_pass_extreme.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    sim.sim_col_pass_extreme.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    big_c = _ntt_primes(N, 2 ** 60 - 2 ** 26 + 1, 2 ** 60 - 2 ** 26 + 2 ** 24, 2)
    small_c = _ntt_primes(N, 2 ** 59, 2 ** 60, 2)
    primes = big_c + small_c
    assert len(primes) == 4 and all(2 ** 60 - p < 2 ** 26 for p in primes)
    arr = (C.c_uint64 * len(primes))(*primes)
    h = sim.sim_params_create_primes(2, N, arr, len(primes), 0)
    assert h
    ctx = oracle.Context(oracle.SCHEME_CKKS, N, primes=primes)
    rng = np.random.default_rng(N)
    try:
        for i, q in enumerate(primes):
            for value in (4 * q - 1, 2 * q, q - 1, 0):
                assert sim.sim_row_pass_extreme(h, i, value) == 0, (hex(q), value)
                assert sim.sim_col_pass_extreme(h, i, value) == 0, (hex(q), value)
            a = rng.integers(0, q, N, dtype=np.uint64)
            f = a.copy()
            sim.sim_ntt_forward(h, i, oracle._p(f))
            assert np.array_equal(f, ctx.ntt(i, a))
            for b in (a, np.full(N, q - 1, dtype=np.uint64)):
                g = ctx.ntt(i, b)
                sim.sim_overflow_reset()
                sim.sim_ntt_inverse(h, i, oracle._p(g))
                assert np.array_equal(g, b)
                assert sim.sim_overflow_reset() == 0
    finally:
        sim.sim_params_destroy(h)
