"""Exact big-integer model of the evaluator operations (TEST INFRASTRUCTURE; second source for the oracle's arithmetic).

Written from the mathematical definitions only — Python integers, CRT composition, exact floors, polynomial products in
Z[X]/(X^N + 1) by Kronecker substitution — with none of the RNS shortcuts the oracle (oracle/he_oracle.c) and the HIP kernels
share (Harvey butterflies, Barrett/Shoup reductions, per-prime correction terms).  What it pins, bit for bit:

  * NTT form        value i of a residue polynomial = a(psi^(2 bitrev(i) + 1)) mod q, psi the SMALLEST primitive 2N-th root
                    (SURVEY.md App. A.2); `ntt_by_definition` evaluates that directly, `ntt` is a recursive split checked
                    against it by the generator.
  * multiply (CKKS) negacyclic products of the coefficient polynomials, (c0, c1, c2) = (a0 b0, a0 b1 + a1 b0, a1 b1).
  * key switching   digits d_j = coefficients of the target under q_j as integers in [0, q_j); S_k = sum_j d_j * key_j[k] in
                    Z_{Q P}[X]/(X^N+1); result_k = floor((S_k + floor(P/2)) / P) mod q_i, S_k the representative in [0, Q P)
                    (what SEAL's mod-down computes: (S - ((S + h) mod P - h)) / P).
  * rescale         floor((x + floor(q_last/2)) / q_last) mod q_i, x the representative in [0, Q_L) (App. A.6).
  * Galois          a(X) -> a(X^g) on coefficients (X^N = -1), then the key switch of the second polynomial (App. A.8).

  * BFV multiply    BEHZ (Bajard-Eynard-Hasan-Zucca) as an INTEGER statement.  The result is not floor(t c1 c2 / Q) of the residues —
                    it depends on the representatives the base conversions pick — but it is a closed integer formula once those
                    are named: x' = (v + Q r) / m~ with v = sum_i [x_i m~ (Q/q_i)^-1]_{q_i} (Q/q_i) (fast base conversion, no
                    correction), r = [-v Q^-1]_{m~} centred, m~ = 2^32 (an exact division: the small Montgomery reduction);
                    D = tensor of the x' over Z[X]/(X^N+1); u = sum_i [(t D)_i (Q/q_i)^-1]_{q_i} (Q/q_i);
                    result = ((t D - u) / Q) mod q_j (exact division again: the fast floor), which the Shenoy-Kumaresan step
                    returns exactly whatever auxiliary base B is used (it only has to be large enough).  `bfv_multiply` below
                    computes that with big integers — no auxiliary primes, no NTT — so it checks the oracle's whole BEHZ pipeline,
                    base choice included, against the algorithm's integer meaning.

Data layouts are SEAL's: ciphertext [size][L][N], key [L digits][2][K][N] in NTT form, key prime K-1 = special prime.
"""
from __future__ import annotations


def bitrev(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def minimal_primitive_root(q: int, two_n: int) -> int:
    """smallest integer that is a primitive two_n-th root of unity mod q (q = 1 mod two_n, two_n a power of two)"""
    assert (q - 1) % two_n == 0
    g = 2
    while True:  # any primitive two_n-th root: x^((q-1)/two_n) for a non-residue x
        r = pow(g, (q - 1) // two_n, q)
        if pow(r, two_n // 2, q) == q - 1:
            break
        g += 1
    best, sq, cur = r, r * r % q, r
    for _ in range(two_n // 2):  # the primitive roots are the odd powers of r
        best = min(best, cur)
        cur = cur * sq % q
    return best


class Ring:
    def __init__(self, N: int, primes: list[int]):
        self.N, self.logn, self.primes = N, N.bit_length() - 1, list(primes)
        self.psi = [minimal_primitive_root(q, 2 * N) for q in primes]
        self._pts = {}

    # ---- NTT form <-> coefficients -------------------------------------------------------------------
    def points(self, i: int) -> list[int]:
        if i not in self._pts:
            q, psi = self.primes[i], self.psi[i]
            self._pts[i] = [pow(psi, 2 * bitrev(k, self.logn) + 1, q) for k in range(self.N)]
        return self._pts[i]

    def ntt_by_definition(self, i: int, coeffs: list[int]) -> list[int]:
        q = self.primes[i]
        out = []
        for x in self.points(i):
            acc = 0
            for c in reversed(coeffs):  # Horner
                acc = (acc * x + c) % q
            out.append(acc)
        return out

    def ntt(self, i: int, coeffs: list[int]) -> list[int]:
        """same values by recursive splitting: a(x) = e(x^2) + x o(x^2); X^N = -1 = psi^N"""
        q, psi, n = self.primes[i], self.psi[i], self.N
        # evaluate at psi^(2k+1) for k in natural order, then permute to bit-reversed order
        def rec(a, w):  # a: coefficients, w: a primitive 2*len(a)-th root; returns [a(w^(2k+1)) for k < len(a)]
            m = len(a)
            if m == 1:
                return [a[0] % q]
            e, o = rec(a[0::2], w * w % q), rec(a[1::2], w * w % q)
            out = [0] * m
            x = w
            w2 = w * w % q
            for k in range(m // 2):
                t = x * o[k] % q
                out[k] = (e[k] + t) % q
                out[k + m // 2] = (e[k] - t) % q  # w^(2(k+m/2)+1) = -w^(2k+1)
                x = x * w2 % q
            return out
        nat = rec(list(coeffs), psi)
        return [nat[bitrev(k, self.logn)] for k in range(n)]

    def intt(self, i: int, values: list[int]) -> list[int]:
        """interpolation: coefficients from the NTT-form values (exact inverse of ntt)"""
        q, n = self.primes[i], self.N
        psi_inv = pow(self.psi[i], q - 2, q)
        nat = [0] * n
        for k in range(n):
            nat[bitrev(k, self.logn)] = values[k]
        # a_j = N^-1 * psi^-j * sum_k y_k * (psi^2)^(-jk): inverse cyclic DFT, then untwist
        w = psi_inv * psi_inv % q
        def rec(y, w):  # cyclic DFT of y with root w
            m = len(y)
            if m == 1:
                return [y[0] % q]
            e, o = rec(y[0::2], w * w % q), rec(y[1::2], w * w % q)
            out = [0] * m
            x = 1
            for k in range(m // 2):
                t = x * o[k] % q
                out[k] = (e[k] + t) % q
                out[k + m // 2] = (e[k] - t) % q
                x = x * w % q
            return out
        d = rec(nat, w)
        ninv = pow(n, q - 2, q)
        out, tw = [], ninv
        for j in range(n):
            out.append(d[j] * tw % q)
            tw = tw * psi_inv % q
        return out

    # ---- exact polynomial arithmetic -----------------------------------------------------------------
    def negacyclic_mul(self, a: list[int], b: list[int], q: int) -> list[int]:
        """a * b in Z_q[X]/(X^N + 1) through one big-integer product (Kronecker substitution)"""
        n = self.N
        w = (2 * q.bit_length() + n.bit_length() + 7) // 8 + 1
        A = int.from_bytes(b"".join(int(x).to_bytes(w, "little") for x in a), "little")
        B = int.from_bytes(b"".join(int(x).to_bytes(w, "little") for x in b), "little")
        raw = (A * B).to_bytes(2 * n * w, "little")
        c = [int.from_bytes(raw[k * w:(k + 1) * w], "little") for k in range(2 * n)]
        return [(c[k] - c[k + n]) % q for k in range(n)]

    def crt(self, idx: list[int]):
        """(Q, composer) for the primes idx: composer(residues) -> the integer in [0, Q)"""
        Q = 1
        for i in idx:
            Q *= self.primes[i]
        terms = []
        for i in idx:
            q = self.primes[i]
            punct = Q // q
            terms.append(punct * pow(punct % q, q - 2, q))
        def compose(res):
            return sum(r * t for r, t in zip(res, terms)) % Q
        return Q, compose

    def galois_coeff(self, a: list[int], g: int, q: int) -> list[int]:
        n, out = self.N, [0] * self.N
        for i, v in enumerate(a):
            e = i * g % (2 * n)
            out[e % n] = (q - v) % q if e >= n else v
        return out


class Model:
    """Evaluator operations on SEAL-layout data given as nested lists of Python ints.  ntt_form: CKKS True, BFV False."""

    def __init__(self, N: int, key_primes: list[int], ntt_form: bool):
        self.R = Ring(N, key_primes)
        self.N, self.K, self.ntt_form = N, len(key_primes), ntt_form

    def to_coeff(self, poly_residues):  # [L][N] data form -> coefficient form
        return [self.R.intt(i, r) for i, r in enumerate(poly_residues)] if self.ntt_form else [list(r) for r in poly_residues]

    def from_coeff(self, poly_residues):
        return [self.R.ntt(i, r) for i, r in enumerate(poly_residues)] if self.ntt_form else [list(r) for r in poly_residues]

    def add(self, a, b):
        return [[[(x + y) % self.R.primes[i] for x, y in zip(pa[i], pb[i])] for i in range(len(pa))] for pa, pb in zip(a, b)]

    def multiply_ckks(self, a, b):
        L = len(a[0])
        ca, cb = [self.to_coeff(p) for p in a], [self.to_coeff(p) for p in b]
        out = [[], [], []]
        for i in range(L):
            q, mul = self.R.primes[i], self.R.negacyclic_mul
            c0 = mul(ca[0][i], cb[0][i], q)
            c1 = [(x + y) % q for x, y in zip(mul(ca[0][i], cb[1][i], q), mul(ca[1][i], cb[0][i], q))]
            c2 = mul(ca[1][i], cb[1][i], q)
            for k, c in enumerate((c0, c1, c2)):
                out[k].append(c)
        return [self.from_coeff(p) for p in out]

    def key_switch_coeff(self, target_coeff, key):
        """target_coeff [L][N] coefficient residues; key [Ltop][2][K][N] NTT form.  Returns the two polynomials to add, as
        coefficient residues [2][L][N]: floor((sum_j d_j key_j[k] + floor(P/2)) / P) mod q_i."""
        L, K, N = len(target_coeff), self.K, self.N
        sp = K - 1
        primes = list(range(L)) + [sp]
        P = self.R.primes[sp]
        QP, compose = self.R.crt(primes)
        out = []
        for k in range(2):
            S = []  # per prime of `primes`: the sum polynomial
            for t in primes:
                p = self.R.primes[t]
                acc = [0] * N
                for j in range(L):
                    d = [v % p for v in target_coeff[j]]            # the digit, an integer polynomial with coefficients in [0, q_j)
                    kc = self.R.intt(t, key[j][k][t])               # the key polynomial under prime t, coefficient form
                    prod = self.R.negacyclic_mul(d, kc, p)
                    acc = [(x + y) % p for x, y in zip(acc, prod)]
                S.append(acc)
            res = [[0] * N for _ in range(L)]
            for n in range(N):
                s = compose([S[m][n] for m in range(len(primes))])
                v = (s + P // 2) // P
                for i in range(L):
                    res[i][n] = v % self.R.primes[i]
            out.append(res)
        return out

    def relinearize(self, ct3, rk):
        L = len(ct3[0])
        ks = self.key_switch_coeff(self.to_coeff(ct3[2]), rk)
        c = [self.to_coeff(ct3[0]), self.to_coeff(ct3[1])]
        res = [[[(c[k][i][n] + ks[k][i][n]) % self.R.primes[i] for n in range(self.N)] for i in range(L)] for k in range(2)]
        return [self.from_coeff(p) for p in res]

    def rescale(self, ct):
        L, N = len(ct[0]), self.N
        Q, compose = self.R.crt(list(range(L)))
        ql = self.R.primes[L - 1]
        out = []
        for p in ct:
            c = self.to_coeff(p)
            res = [[0] * N for _ in range(L - 1)]
            for n in range(N):
                x = compose([c[i][n] for i in range(L)])
                v = (x + ql // 2) // ql
                for i in range(L - 1):
                    res[i][n] = v % self.R.primes[i]
            out.append(self.from_coeff(res))
        return out

    def apply_galois(self, ct, g, gkey):
        L = len(ct[0])
        c = [self.to_coeff(ct[0]), self.to_coeff(ct[1])]
        r = [[self.R.galois_coeff(c[k][i], g, self.R.primes[i]) for i in range(L)] for k in range(2)]
        ks = self.key_switch_coeff(r[1], gkey)
        res0 = [[(r[0][i][n] + ks[0][i][n]) % self.R.primes[i] for n in range(self.N)] for i in range(L)]
        return [self.from_coeff(res0), self.from_coeff(ks[1])]

    # ---- BFV ct x ct multiply: the BEHZ pipeline as integer arithmetic (see the module docstring) ------------------------
    def bfv_multiply(self, a, b, t: int):
        assert not self.ntt_form
        L, N = len(a[0]), self.N
        idx = list(range(L))
        primes = [self.R.primes[i] for i in idx]
        Q = 1
        for q in primes:
            Q *= q
        MT = 1 << 32
        punct = [Q // q for q in primes]
        inv_punct = [pow(pq % q, q - 2, q) for pq, q in zip(punct, primes)]
        Qinv_mt = pow(Q, -1, MT)

        def representative(poly):  # [L][N] residues -> integer coefficients x' = SmMRq(FastBconv_mtilde(x))
            out = []
            for n in range(N):
                v = sum((poly[i][n] * MT % primes[i]) * inv_punct[i] % primes[i] * punct[i] for i in idx)
                r = (-v * Qinv_mt) % MT
                if r >= MT // 2:
                    r -= MT
                num = v + Q * r
                assert num % MT == 0
                out.append(num // MT)
            return out

        xa, xb = [representative(p) for p in a], [representative(p) for p in b]
        bound = max(max(abs(v) for v in p) for p in xa + xb)
        M = 1 << ((2 * bound * bound * N).bit_length() + 2)  # products computed mod M and re-centred: exact over Z

        def zmul(x, y):
            z = self.R.negacyclic_mul([v % M for v in x], [v % M for v in y], M)
            return [v - M if v >= M // 2 else v for v in z]

        d0 = zmul(xa[0], xb[0])
        d1 = [p + q for p, q in zip(zmul(xa[0], xb[1]), zmul(xa[1], xb[0]))]
        d2 = zmul(xa[1], xb[1])
        out = []
        for d in (d0, d1, d2):
            res = [[0] * N for _ in idx]
            for n in range(N):
                td = t * d[n]
                u = sum((td % primes[i]) * inv_punct[i] % primes[i] * punct[i] for i in idx)
                assert (td - u) % Q == 0
                f = (td - u) // Q
                for i in idx:
                    res[i][n] = f % primes[i]
            out.append(res)
        return out
