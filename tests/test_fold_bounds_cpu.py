"""The inequalities the fold form of the u64 engine rests on (csrc/modarith.h: q = 2^60 - c, c < 2^26), in plain Python integers -- the same
bounds the lane simulator flags at run time (tests/test_lane_sim.py), here as arithmetic anyone can read: every lazy value fits 64 bits, every
offset dominates what is subtracted from it, every reduction lands where its consumer expects it."""
import random

import pytest

CS = [1, 2 ** 14 - 1, 2 ** 18 - 1, 0x7BFFFF, 2 ** 24, 2 ** 26 - 1]  # the reference's 60-bit primes have c = 2^14-1, 2^18-1, 0x7bffff; the form admits c < 2^26
U64 = 2 ** 64


def fold_sum(x, w, w2):
    x0, x1 = x & 0xFFFFFFFF, x >> 32
    return x0 * w + x1 * w2  # what the four 32 x 32 -> 64 multiply-adds form


def tight(x, w, w2, c):
    s = fold_sum(x, w, w2)
    return (s & (2 ** 60 - 1)) + (s >> 61) * 2 * c + ((s >> 60) & 1) * c


def wide(x, w, w2, c):
    s = fold_sum(x, w, w2)
    return (s & (2 ** 61 - 1)) + (s >> 61) * 2 * c


def red2q(x, c):
    return (x & (2 ** 61 - 1)) + (x >> 61) * 2 * c


@pytest.mark.parametrize("c", CS)
def test_fold_products_and_their_bounds(c):
    q = 2 ** 60 - c
    rng = random.Random(c)
    worst_t = worst_w = 0
    xs = [0, 1, q - 1, 4 * q - 1, 16 * q - 1, U64 - 1, 0xFFFFFFFF, 0xFFFFFFFF00000000] + [rng.getrandbits(64) for _ in range(2000)]
    ws = [0, 1, q - 1, q // 2, 0xFFFFFFFF, 2 ** 59 + 12345] + [rng.randrange(q) for _ in range(20)]
    for x in xs:
        for w in ws:
            w2 = (w << 32) % q
            s = fold_sum(x, w, w2)
            assert s < 2 ** 93 and (s >> 61) < 2 ** 32          # S >> 61 is a 32-bit multiplier operand
            t, v = tight(x, w, w2, c), wide(x, w, w2, c)
            assert t % q == v % q == (x * w) % q
            worst_t, worst_w = max(worst_t, t), max(worst_w, v)
    assert worst_t < 2 ** 60 + 2 ** 33 * c < 2 * q                  # the tight product replaces Shoup's "below 2q" everywhere
    assert worst_w < 2 ** 61 + 2 ** 33 * c <= 3 * q                 # the wide product: the offsets of 3q dominate it


@pytest.mark.parametrize("c", CS)
def test_lazy_ranges_fit_64_bits(c):
    q = 2 ** 60 - c
    W = 2 ** 61 + 2 ** 33 * c - 1          # largest wide product
    R = 2 ** 61 + 14 * c                   # fold_red2q of any 64-bit value stays below this
    assert all(red2q(x, c) < R and red2q(x, c) % q == x % q for x in (0, U64 - 1, 16 * q - 1, 2 ** 61, 2 ** 63 + 5))
    # forward row pass: in below 4q; a stage takes values below B to values below B + 3q (X + v and X + 3q - v)
    assert W <= 3 * q
    b = 4 * q - 1
    for _ in range(4):
        b += 3 * q
    assert b < U64                         # phase A: four stages
    b = R - 1
    for _ in range(4):
        b += 3 * q
    assert b < U64 and b < 14 * q + 16 * c  # phase B after lazy_reduce
    b = R - 1 + 2 * 3 * q
    assert b < 8 * q + 16 * c              # phase C: what the key products read
    # forward column pass: at most four stages from below 4q, lazy_reduce before a fifth, lazy_reduce at the end: below 2q + 16c < 4q
    assert R <= 2 * q + 16 * c < 4 * q
    # key products: a sum below R takes five wide products
    assert R - 1 + 5 * W < U64
    # inverse butterflies: values below B = 2^61 + 2^33 c
    B = 2 ** 61 + 2 ** 33 * c
    assert 2 * B < U64 and B <= 3 * q and B + 3 * q < U64 and R <= B
    # last inverse stage: tight products, what leaves a pass is below 2q
    assert 2 ** 60 + 2 ** 33 * c < 2 * q


def test_the_rule_of_the_reference_produces_such_primes():
    """{60, b ..., 60}: the largest primes 1 (mod 2N) below 2^60 (seal_context.cpp:79-82) -- the first seventeen of them have c < 2^26 at
    every ring size the reference uses, so a chain of up to sixteen 60-bit data primes plus the special prime takes the fold form."""
    sympy = pytest.importorskip("sympy")
    for N in (4096, 8192, 16384, 32768):
        found, v = [], (2 ** 60 - 2) // (2 * N) * (2 * N) + 1
        while len(found) < 17:
            if sympy.isprime(v):
                found.append(v)
            v -= 2 * N
        assert all(2 ** 60 - p < 2 ** 26 for p in found), (N, [hex(2 ** 60 - p) for p in found])
