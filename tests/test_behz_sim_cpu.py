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


def _lists(n, gs, b1):
    """The distinct-operand lists of a batch as DeviceContext::bfv_multiply3 sizes them (he355_api.hip)."""
    gsz = min(gs, n)
    G = 1 if gs >= n else (n + gs - 1) // gs
    I, J = (gsz + b1 - 1) // b1, min(b1, gsz)
    return G, I, J, G * I


@pytest.mark.parametrize("name,n,ix", [
    # a_base, b_base, gs, b1, a_sg, a_si, b_sg, b_sj
    ("outer_3x2", 6, (0, 0, 2 ** 64 - 1, 2, 0, 1, 0, 1)),
    ("outer_ragged", 10, (5, 7, 2 ** 64 - 1, 4, 0, 1, 0, 1)),          # last row of the outer product incomplete
    ("matrix_10x9x8", 720, (0, 0, 80, 8, 10, 1, 8, 1)),                # the CipherBatchAxis layout: a(i,k) at k*10+i, b(k,j) at k*8+j
    ("matrix_row_major", 30, (2, 3, 6, 3, 1, 5, 3, 1)),                # a(i,k) at i*5+k, b(k,j) at k*3+j, bases > 0
])
def test_transformed_once_lists_address_each_results_operands(sim, name, n, ix):
    """The invariant the hoisted BFV multiply rests on: the list item a result's ordinal points at was extended from that result's
    operand (behz_src_ct(lists, ord(r)) == idx(r)), ordinals stay inside the lists, and every list item serves at least one result."""
    sim.sim_behz_src_map.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    a_base, b_base, gs, b1, a_sg, a_si, b_sg, b_sj = ix
    G, I, J, na = _lists(n, gs, b1)
    nb = G * J
    ixa = arr(ix)
    used = set()
    for r in range(n):
        out = (C.c_uint64 * 5)()
        sim.sim_behz_src_map(ixa, I, J, na, r, 0, out)
        ia, ib, oa, ob = (int(out[k]) for k in range(4))
        g, rr = (r // gs, r % gs) if gs < 2 ** 63 else (0, r)
        assert ia == a_base + g * a_sg + (rr // b1) * a_si and ib == b_base + g * b_sg + (rr % b1) * b_sj
        assert 0 <= oa < na <= ob < na + nb
        for item, want, is_b in ((oa, ia, False), (ob, ib, True)):
            sim.sim_behz_src_map(ixa, I, J, na, r, item, out)
            src = int(out[4])
            assert (src >= 2 ** 62) == is_b and (src - 2 ** 62 if is_b else src) == want, (name, r, item)
        used.update((oa, ob))
    if n % min(gs, n) == 0 and min(gs, n) % b1 == 0:
        assert used == set(range(na + nb))  # complete groups: no list item is extended in vain
    assert na + nb <= n  # (the shapes here are ones the product hoists)
