"""The auxiliary base of the BEHZ multiply (csrc/he_params.h, Params::aux; csrc/he_params.cpp, Params::behz_base_suffices).

SEAL's RNSTool takes 61-bit auxiliary primes; the device takes 46-bit ones (fp64 engine) because the product is the same integer for
any base that is large enough for the Shenoy-Kumaresan step.  These tests hold the host-side rule to an independent statement of
the bound in Python integers / Fractions, and run the Shenoy-Kumaresan step itself -- as integer arithmetic, on the base the
library reports -- at the extremes of its input range.  (The device code is held to the oracle, which restates SEAL with SEAL's
base, on extreme operands in tests/test_gpu_parity_bfv.py::test_bfv_multiply_is_independent_of_the_auxiliary_base.)"""
import importlib
import random
from fractions import Fraction

import pytest

PARAMS = [
    (8192, [60, 40, 60], 20),              # reference defaults, BFV element-wise / dot product
    (8192, [60, 40, 40, 60], 20),          # reference defaults, BFV matrix workloads
    (32768, [60, 40, 40, 60], 20),         # BASELINE configs[4]
    (16384, [60, 40, 40, 40, 60], 20),
    (4096, [60, 60], 20),
    (2048, [60, 60, 60, 60, 60], 31),      # largest plain modulus the base accepts, 60-bit data primes only
    (2048, [50, 40, 40, 45, 40, 60, 60], 22),
    (1024, [46, 46, 46], 16),              # coefficient moduli of the auxiliary primes' own size: the search must step over them
    (32768, [60] + [45] * 16 + [60], 20),  # 17 data primes
]


@pytest.fixture(scope="module")
def be():
    return importlib.import_module("reference-seal-backend_amd")


def is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def v_max(t, N, qs):
    """Largest |V| the fast floor can hand to the Shenoy-Kumaresan step: |Y| <= Q (1/2 + L / 2^32) for the extended operands,
    |Z| <= 2 N Y^2 for c1 = a0 b1 + a1 b0, V = (t Z - W) / Q with 0 <= W < L Q."""
    L = len(qs)
    Q = 1
    for q in qs:
        Q *= q
    y = Q * (Fraction(1, 2) + Fraction(L, 2 ** 32))
    return t * 2 * N * y * y / Q + L


def sk_suffices(vmax, base):
    msk, B = base[0], 1
    for b in base[1:]:
        B *= b
    return len(base) - 1 + vmax / B <= Fraction(msk - 1, 2)


def shenoy_kumaresan(V, base, qs):
    """RNSTool::fastbconv_sk as integers: from V mod b_i and V mod m_sk to V mod q_j."""
    msk, bs = base[0], base[1:]
    B = 1
    for b in bs:
        B *= b
    tmp = [(V % b) * pow(B // b, -1, b) % b for b in bs]
    alpha = (sum(x * ((B // b) % msk) for x, b in zip(tmp, bs)) - V % msk) * pow(B, -1, msk) % msk
    out = []
    for q in qs:
        conv = sum(x * ((B // b) % q) for x, b in zip(tmp, bs)) % q
        if alpha > msk >> 1:
            out.append((conv + (msk - alpha) * (B % q)) % q)
        else:
            out.append((conv - alpha * (B % q)) % q)
    return out


@pytest.mark.parametrize("N,bits,pb", PARAMS)
def test_auxiliary_base_meets_the_shenoy_kumaresan_bound_at_every_level(be, monkeypatch, N, bits, pb):
    monkeypatch.delenv("HE355_BEHZ_BASE", raising=False)
    ctx = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False)
    rnd = random.Random(N + len(bits))
    full = ctx.bfv_aux_base(ctx.L)
    for L in range(1, ctx.L + 1):
        base = ctx.bfv_aux_base(L)
        assert base == full[:len(base)]                                   # one prime list, a prefix per level
        assert len(set(base)) == len(base) and not set(base) & set(ctx.moduli)
        assert all(p.bit_length() == 46 and p % (2 * N) == 1 and is_prime(p) for p in base)
        qs = ctx.moduli[:L]
        vmax = v_max(ctx.t, N, qs)
        assert sk_suffices(vmax, base), (L, base)
        if len(base) > 2:
            assert not sk_suffices(vmax, base[:-1]), "the rule took more primes than the bound asks for"
        # the step itself, at the ends of its input range and inside it
        top = int(vmax)
        for V in [top, -top, top - 1, -top + 1, 0, 1, -1] + [rnd.randint(-top, top) for _ in range(8)]:
            assert shenoy_kumaresan(V, base, qs) == [V % q for q in qs], (L, V)
        # ... and the test has teeth: one prime short, the largest negative input is decoded wrongly
        if len(base) > 2 and not sk_suffices(vmax, base[:-1]):
            assert shenoy_kumaresan(-top, base[:-1], qs) != [(-top) % q for q in qs]
    # never more auxiliary residues than SEAL's |q| + 1 for the reference's own parameter sets
    if bits in ([60, 40, 60], [60, 40, 40, 60], [60, 40, 40, 40, 60]):
        assert len(full) == ctx.L + 1
    ctx.close()


def test_seal_base_is_selectable_and_is_seals(be, monkeypatch):
    """HE355_BEHZ_BASE=seal: m_sk, B_0.. = primes 0, 2, 3, .. of get_primes(2N, 61, |q| + 2) (prime 1 is gamma)."""
    monkeypatch.setenv("HE355_BEHZ_BASE", "seal")
    N = 4096
    ctx = be.Context(be.SCHEME_BFV, N, bit_sizes=[60, 40, 40, 60], plain_bits=20, sec128=False)
    found, v = [], ((1 << 61) - 1) // (2 * N) * (2 * N) + 1
    while len(found) < ctx.L + 2:
        if is_prime(v):
            found.append(v)
        v -= 2 * N
    assert ctx.bfv_aux_base(ctx.L) == [found[0]] + found[2:]
    assert ctx.bfv_aux_base(1) == [found[0], found[2]]
    ctx.close()
    monkeypatch.delenv("HE355_BEHZ_BASE")
    ckks = be.Context(be.SCHEME_CKKS, N, bit_sizes=[60, 40, 60], sec128=False)
    assert ckks.bfv_aux_base(1) == []
    ckks.close()
