"""Seeded inputs of the exact-model fixtures (tests/golden/exact_vectors.json): shared by the generator
(make_exact_vectors.py, pure Python integers) and by the tests, so both sides work on identical data.
Values are uniform residues from a splitmix64 counter stream: value(tag, index) = floor(rand64 * q / 2^64)."""
from __future__ import annotations

M64 = (1 << 64) - 1

CASES = [
    # name, scheme, N, key-level bit sizes (the last one is the special prime), what to compute
    dict(name="ckks_n1024_60_40_60", scheme="ckks", N=1024, bits=[60, 40, 60], seed=0xE0A1),   # u64 / fp64 / u64 engines
    dict(name="ckks_n1024_50_45_45_50", scheme="ckks", N=1024, bits=[50, 45, 45, 50], seed=0xE0A2),  # two fp64-engine data primes
    dict(name="ckks_n2048_60_45_45_60", scheme="ckks", N=2048, bits=[60, 45, 45, 60], seed=0xE0A3),  # N1 = 2: column pass + row pass
    dict(name="bfv_n1024_60_40_60", scheme="bfv", N=1024, bits=[60, 40, 60], seed=0xE0B1),
    dict(name="bfv_n2048_60_40_40_60", scheme="bfv", N=2048, bits=[60, 40, 40, 60], seed=0xE0B2),  # three data primes, N1 = 2
]


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def uniform_poly(seed: int, tag: int, q: int, N: int) -> list[int]:
    base = splitmix64(seed ^ (tag * 0x9E3779B97F4A7C15 & M64))
    return [(splitmix64(base ^ splitmix64(i)) * q) >> 64 for i in range(N)]


def ciphertext(seed: int, tag: int, primes: list[int], L: int, size: int, N: int):
    """[size][L][N] uniform residues"""
    return [[uniform_poly(seed, tag * 1000 + k * 100 + i, primes[i], N) for i in range(L)] for k in range(size)]


def kswitch_key(seed: int, tag: int, primes: list[int], Ltop: int, N: int):
    """[Ltop][2][K][N] uniform residues (a uniformly random key exercises the same arithmetic as a real one)"""
    K = len(primes)
    return [[[uniform_poly(seed, tag * 100000 + j * 1000 + k * 100 + t, primes[t], N) for t in range(K)] for k in range(2)] for j in range(Ltop)]
