"""The product's per-coefficient BEHZ arithmetic on the CPU (tests/csim/sim_behz.cpp runs csrc/behz_core.h -- the functions the HIP kernels
compile -- on the constants the product uploads, Params::behz_host) against exact integer arithmetic in Python:

* steps (1)-(2), base extension with the Montgomery correction: out_j = ((X + Q r) / 2^32) mod p_j with X the fast conversion of
  2^32 x and r = -X Q^-1 mod 2^32 centred;
* steps (6)-(8), times t, fast floor, Shenoy-Kumaresan: out_i = ((t D - W) / Q) mod q_i with W the fast conversion of t D mod Q --
  in the integer version AND in the fp64-engine version the 46-bit auxiliary base runs, for products D up to the largest the
  tensor step can hand over (|D| <= 2 N Y^2, Y = Q (1/2 + L / 2^32)): the ends of the Shenoy-Kumaresan bound (he_params.cpp,
  behz_base_suffices) are inside the tested range.
No GPU, no oracle: the expected values are Python integers."""
import ctypes as C
import os
import random
import subprocess
from fractions import Fraction

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = [
    (8192, [60, 40, 60], 20),              # <4, 6> instantiation, L = 2 = nB
    (8192, [60, 40, 40, 60], 20),
    (32768, [60, 40, 40, 60], 20),         # BASELINE configs[4]
    (16384, [60, 40, 40, 40, 60], 20),
    (2048, [60, 60, 60, 60, 60], 31),      # four 60-bit data primes, the largest plain modulus: six auxiliary primes
    (2048, [50, 40, 40, 45, 40, 60, 60], 22),  # <16, 24> instantiation, both engines among the data primes
    (1024, [46, 46, 46], 16),
]


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "csim"), "-s"], check=True)
    L = C.CDLL(os.path.join(HERE, "csim", "_build", "libcsim.so"))
    u64p = C.POINTER(C.c_uint64)
    L.sim_behz_create.restype = C.c_void_p
    L.sim_behz_create.argtypes = [C.c_size_t, C.POINTER(C.c_int), C.c_size_t, C.c_int]
    L.sim_behz_destroy.argtypes = [C.c_void_p]
    L.sim_behz_levels.restype = C.c_size_t
    L.sim_behz_levels.argtypes = [C.c_void_p]
    L.sim_behz_q.restype = C.c_uint64
    L.sim_behz_q.argtypes = [C.c_void_p, C.c_size_t]
    L.sim_behz_t.restype = C.c_uint64
    L.sim_behz_t.argtypes = [C.c_void_p]
    L.sim_behz_base.restype = C.c_size_t
    L.sim_behz_base.argtypes = [C.c_void_p, C.c_int, u64p]
    L.sim_behz_f64aux.argtypes = [C.c_void_p, C.c_int]
    L.sim_behz_extend.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
    L.sim_behz_floor.argtypes = [C.c_void_p, C.c_int, u64p, u64p, u64p, C.c_int]
    return L


def arr(vals):
    return (C.c_uint64 * len(vals))(*[int(v) for v in vals])


def expected_extension(x, qs, bsk):
    MT = 1 << 32
    Q = 1
    for q in qs:
        Q *= q
    X = sum(((x % q) * MT % q) * pow(Q // q, -1, q) % q * (Q // q) for q in qs)
    r = (-X * pow(Q, -1, MT)) % MT
    if r >= MT // 2:
        r -= MT
    assert (X + Q * r) % MT == 0
    Y = (X + Q * r) // MT
    return [Y % p for p in bsk], Y


def expected_floor(D, t, qs):
    Q = 1
    for q in qs:
        Q *= q
    W = sum(((t * D) % q) * pow(Q // q, -1, q) % q * (Q // q) for q in qs)
    assert (t * D - W) % Q == 0
    V = (t * D - W) // Q
    return [V % q for q in qs]


@pytest.mark.parametrize("base", ["device", "seal"])
@pytest.mark.parametrize("N,bits,pb", PARAMS)
def test_behz_coefficient_arithmetic_equals_integer_arithmetic(sim, monkeypatch, base, N, bits, pb):
    if base == "seal":
        monkeypatch.setenv("HE355_BEHZ_BASE", "seal")
    else:
        monkeypatch.delenv("HE355_BEHZ_BASE", raising=False)
    h = sim.sim_behz_create(N, (C.c_int * len(bits))(*bits), len(bits), pb)
    assert h
    rnd = random.Random(N * 31 + len(bits) + (base == "seal"))
    t = int(sim.sim_behz_t(h))
    try:
        for L in range(1, int(sim.sim_behz_levels(h)) + 1):
            qs = [int(sim.sim_behz_q(h, i)) for i in range(L)]
            buf = (C.c_uint64 * 64)()
            n = sim.sim_behz_base(h, L, buf)
            msk, bs = int(buf[0]), [int(buf[i]) for i in range(1, n)]
            bsk = bs + [msk]  # the kernels' residue order: B_0 .. B_{nB-1}, m_sk
            f64aux = sim.sim_behz_f64aux(h, L)
            assert f64aux == (1 if base == "device" else 0)
            Q = 1
            for q in qs:
                Q *= q
            # steps (1)-(2)
            ymax = 0
            for x in [0, 1, Q - 1, Q // 2, Q // 2 + 1] + [rnd.randrange(Q) for _ in range(24)]:
                out = (C.c_uint64 * len(bsk))()
                assert sim.sim_behz_extend(h, L, arr([x % q for q in qs]), out) == 0
                want, Y = expected_extension(x, qs, bsk)
                assert [int(v) for v in out] == want, (L, x)
                assert abs(Y) <= Q * (Fraction(1, 2) + Fraction(L, 2 ** 32))  # the bound the base is sized for
                ymax = max(ymax, abs(Y))
            # steps (6)-(8): products up to the largest the tensor step can produce
            ybound = int(Q * (Fraction(1, 2) + Fraction(L, 2 ** 32)))
            dmax = 2 * N * ybound * ybound
            ds_list = [dmax, -dmax, dmax - 1, -dmax + 1, 0, 1, -1, Q, -Q] + [rnd.randint(-dmax, dmax) for _ in range(24)]
            for D in ds_list:
                want = expected_floor(D, t, qs)
                for variant in ([0, 1] if f64aux else [0]):
                    out = (C.c_uint64 * L)()
                    assert sim.sim_behz_floor(h, L, arr([D % q for q in qs]), arr([D % p for p in bsk]), out, variant) == 0
                    assert [int(v) for v in out] == want, (L, D, variant)
    finally:
        sim.sim_behz_destroy(h)
