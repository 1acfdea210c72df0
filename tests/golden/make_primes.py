"""Regenerates tests/golden/primes.json with sympy (independent of the oracle and of the product).

Rule restated: SEAL v3.7.2 util/numth.cpp get_primes(factor=2N, bit_size, count) — descending primes
= 1 (mod 2N) below 2^bit_size — and modulus.cpp CoeffModulus::Create, which serves each requested slot from
the BACK of the per-bit-size list.  The reference builds its chain as {60, bits x (depth-1), 60}
(/root/reference/src/engine/seal_context.cpp:79-82,107-110) and its BFV plain modulus with
PlainModulus::Batching(N, bits) (seal_context.cpp:118).  The 60-bit and batching values below agree with
SURVEY.md Appendix B, and 0xffffffffffc0001 / 1032193 are the constants SEAL is publicly known to use.
"""
import json
import os
import sympy


def get_primes(factor, bits, count):
    v = ((1 << bits) - 1) // factor * factor + 1
    lo = 1 << (bits - 1)
    out = []
    while len(out) < count and v > lo:
        if sympy.isprime(v):
            out.append(v)
        v -= factor
    assert len(out) == count
    return out


def coeff_modulus_create(N, bit_sizes):
    lists = {}
    for b in set(bit_sizes):
        lists[b] = get_primes(2 * N, b, bit_sizes.count(b))
    return [lists[b].pop() for b in bit_sizes]


def main():
    doc = {"get_primes": [], "chains": [], "batching": [], "aux61": []}
    for N, bits, count in [(32768, 60, 2), (32768, 45, 15), (32768, 50, 15), (16384, 60, 2), (16384, 45, 7),
                           (8192, 60, 2), (8192, 45, 1), (8192, 40, 1), (4096, 36, 2)]:
        doc["get_primes"].append({"N": N, "bits": bits, "primes": [hex(p) for p in get_primes(2 * N, bits, count)]})
    for N, depth, bits in [(8192, 2, 45), (8192, 2, 40), (8192, 3, 40), (16384, 8, 45), (32768, 16, 45), (32768, 16, 50)]:
        bs = [60] + [bits] * (depth - 1) + [60]
        doc["chains"].append({"N": N, "depth": depth, "bits": bits, "bit_sizes": bs,
                              "primes": [hex(p) for p in coeff_modulus_create(N, bs)]})
    for N in (4096, 8192, 16384, 32768):
        doc["batching"].append({"N": N, "bits": 20, "t": get_primes(2 * N, 20, 1)[0]})
    for N, cnt in [(8192, 4), (32768, 5)]:
        doc["aux61"].append({"N": N, "primes": [hex(p) for p in get_primes(2 * N, 61, cnt)]})
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "primes.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
