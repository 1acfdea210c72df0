#!/usr/bin/env python3
"""Randomised differential run of the HIP path against the CPU oracle (test tool, GPU box): random ring sizes, prime chains (mixed
fp64-/u64-engine primes, forced-u64 contexts), levels, batch sizes on both sides of the latency-shape boundary, chunk sizes with
ragged tails; CKKS multiply -> relinearize (-> rescale), relinearize of size-3 ciphertexts, rotations and rotate_add; BFV (30 % of the
cases) BEHZ multiply, relinearize and a row or column rotation; and for both schemes the rotation chains of round 4 -- he355_rotate_sum
(NAF-prefix trie, walked node by node or level by level with grouped key switches, now and then at a batch that sums the level inside k_k3), he355_rotate_each, he355_accumulate -- every result
compared bit for bit.  usage: tools/fuzz_parity.py <seconds> [seed]   (prints one line per case; exit 1 on the first mismatch)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def chains(be, g, o, rng, L, n, a, da):
    """rotate_sum / rotate_each / accumulate against the oracle's rotate_internal restatement; returns (ok, what)"""
    N = g.N
    keys = {}
    for k in range(5):  # +-1 .. +-16: every NAF term of a step in [-15, 15]
        for st in (1 << k, -(1 << k)):
            e = o.galois_elt(st)
            keys[e] = o.random_kswitch_key(rng)
            g.set_galois_key(e, keys[e])
    walk = bool(rng.random() < 0.7)
    g.set_level_walk(walk)
    what = [f"walk{int(walk)}"]
    steps = [int(x) for x in rng.integers(-15, 16, int(rng.integers(1, 9)))]
    out = g.alloc(n * 2 * L * N)
    g.rotate_sum(L, n, da, steps, out)
    got = out.download((n, 2, L, N))
    ok = True
    for r in range(n):
        want = a[r].copy()
        for st in steps:
            want = o.add(want, o.rotate(a[r], st, keys) if st else a[r])
        ok = ok and np.array_equal(got[r], want)
    what.append(f"rotsum{len(steps)}")
    # now and then a batch whose level walk sums inside k_k3 (KsGroups::sum_out: groups of a multiple of eight ciphertexts, at least 512
    # blocks in the data primes' launch): the input repeated to that size, a sample of the results against the rows checked above
    n1 = max(1, N // 1024)
    big = 8 * -(-512 // (8 * L * n1)) * 8
    if ok and walk and n >= 1 and big * 2 * L * N * 8 <= (1 << 28) and rng.random() < 0.25:
        reps = -(-big // n)
        tiled = np.concatenate([a] * reps)[:big]
        dbig, obig = g.to_device(tiled), g.alloc(big * 2 * L * N)
        g.rotate_sum(L, big, dbig, steps, obig)
        gotb = obig.download((big, 2, L, N))
        ok = all(np.array_equal(gotb[r], got[r % n]) for r in range(big))
        what.append(f"levelsum{big}")
    if ok and g.scheme == be.SCHEME_CKKS:
        each = [int(x) for x in rng.integers(-15, 16, n)]
        g.rotate_each(L, n, da, each, out)
        got = out.download((n, 2, L, N))
        ok = all(np.array_equal(got[r], o.rotate(a[r], each[r], keys) if each[r] else a[r]) for r in range(n))
        what.append("roteach")
    if ok:
        count = int(rng.integers(2, 17))
        acc, tmp = g.to_device(a), g.alloc(n * 2 * L * N)
        g.accumulate(L, n, acc, count, tmp)
        got = acc.download((n, 2, L, N))
        ok = all(np.array_equal(got[r], o.accumulate(a[r], count, keys)) for r in range(n))
        what.append(f"acc{count}")
    g.set_level_walk(True)
    return ok, what


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    be = importlib.import_module("reference-seal-backend_amd")
    import oracle  # noqa: E402  (the checker; test infrastructure)
    oracle.build()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    case = 0
    while time.time() < t_end:
        case += 1
        if rng.random() < 0.3:  # BFV: BEHZ multiply, relinearize, a row / column rotation (also added to another ciphertext)
            N = int(rng.choice([1024, 2048, 4096, 8192, 16384], p=[0.25, 0.25, 0.2, 0.15, 0.15]))
            K = int(rng.integers(2, 6))
            bits = [int(x) for x in rng.integers(35, 61, K)]
            if rng.random() < 0.5:  # the reference's shape {60, small ..., 60}: every u64-engine prime is 2^60 - c, the context takes the fold build
                bits = [60 if (i in (0, K - 1) or rng.random() < 0.2) else int(rng.integers(35, 47)) for i in range(K)]
            pb = int(rng.integers(16, 23))
            seal_base = bool(rng.random() < 0.2)  # SEAL's 61-bit auxiliary base instead of the device's 46-bit one (same bits out)
            if seal_base:
                os.environ["HE355_BEHZ_BASE"] = "seal"
            try:
                try:
                    g = be.Context(be.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False, device=0)
                except be.HE355Error as e:
                    print(f"case {case}: BFV N={N} bits={bits} skipped ({str(e)[:60]})", flush=True)
                    continue
            finally:
                os.environ.pop("HE355_BEHZ_BASE", None)
            o = oracle.Context(oracle.SCHEME_BFV, N, bit_sizes=bits, plain_bits=pb, sec128=False)
            assert g.moduli == o.moduli and g.t == o.t
            L, n = g.L, int(rng.choice([1, 2, 3, 5, 9]))
            g.set_chunk(int(rng.choice([2, 4, 256])))
            a = np.stack([o.random_poly(rng, L, 2) for _ in range(n)])
            b = np.stack([o.random_poly(rng, L, 2) for _ in range(n)])
            da, db = g.to_device(a), g.to_device(b)
            c3 = g.alloc(n * 3 * L * N)
            g.bfv_multiply(L, n, da, db, be.Context.pairwise(), c3)
            got3 = c3.download((n, 3, L, N))
            ok = all(np.array_equal(got3[r], o.bfv_multiply(a[r], b[r])) for r in range(n))
            shape = "pairwise"
            if ok and n >= 2:  # outer product (operands transformed once each where that pays) with a ragged last row
                nres = int(rng.integers(n, n * n + 1))
                co = g.alloc(nres * 3 * L * N)
                g.bfv_multiply(L, nres, da, db, be.Context.outer(0, (nres + n - 1) // n, 0, n), co)
                goto = co.download((nres, 3, L, N))
                ok = all(np.array_equal(goto[r], o.bfv_multiply(a[r // n], b[r % n])) for r in range(nres))
                shape += f"+outer{nres}"
                co.free()
            if ok:
                rk = o.random_kswitch_key(rng)
                g.set_relin_key(rk)
                out = g.alloc(n * 2 * L * N)
                g.relinearize(L, n, c3, out)
                got = out.download((n, 2, L, N))
                ok = all(np.array_equal(got[r], o.relinearize(got3[r], rk)) for r in range(n))
            if ok and n >= 2 and N <= 2048:  # the matrix product over an inner index: out(i, j) = sum_k relin(a(i, k) * b(k, j))
                rows, inner = int(rng.integers(1, n + 1)), min(n, int(rng.integers(1, 4)))
                cols = max(1, n // inner)
                rows = min(rows, max(1, n // inner))
                outm = g.alloc(rows * cols * 2 * L * N)
                # a(i, k) at k * rows + i (needs inner * rows <= n), b(k, j) at k * cols + j
                g.bfv_multiply_relin_accumulate(L, rows, cols, inner, da, 1, rows, db, cols, 1, outm)
                gotm = outm.download((rows * cols, 2, L, N))
                mods = [int(q) for q in o.moduli[:L]]
                for i in range(rows):
                    for j in range(cols):
                        acc = np.zeros((2, L, N), dtype=object)
                        for k in range(inner):
                            acc = acc + o.relinearize(o.bfv_multiply(a[k * rows + i], b[k * cols + j]), rk).astype(object)
                        want = np.stack([[acc[pp][l] % mods[l] for l in range(L)] for pp in range(2)]).astype(np.uint64)
                        ok = ok and np.array_equal(gotm[i * cols + j], want)
                shape += f"+mat{rows}x{inner}x{cols}"
                outm.free()
            if ok:
                elt = 2 * N - 1 if rng.random() < 0.3 else g.galois_elt(int(rng.choice([1, 2, -1, 4])))
                gk = o.random_kswitch_key(rng)
                g.set_galois_key(elt, gk)
                rot = g.alloc(n * 2 * L * N)
                g.apply_galois(L, n, da, elt, rot)
                gotr = rot.download((n, 2, L, N))
                ok = all(np.array_equal(gotr[r], o.apply_galois(a[r], elt, gk)) for r in range(n))
            extra = []
            if ok and g.K >= 2:
                g.set_latency_max(int(rng.choice([0, 8])))
                ok, extra = chains(be, g, o, rng, L, n, a, da)
            print(f"case {case}: BFV N={N} bits={bits} t_bits={pb}{' seal-base' if seal_base else ''} L={L} n={n} multiply[{shape}]+relin+galois+{'+'.join(extra)} {'ok' if ok else 'MISMATCH'}", flush=True)
            g.close()
            if not ok:
                return 1
            continue
        N = int(rng.choice([1024, 2048, 4096, 8192, 16384], p=[0.3, 0.25, 0.2, 0.15, 0.1]))
        n_data = int(rng.integers(1, 6 if N <= 4096 else 4))
        bits = [int(rng.choice([36, 40, 44, 45, 46, 47, 50, 52, 55, 60])) for _ in range(n_data)]
        if rng.random() < 0.5:  # fold build of the u64 engine: only 60-bit primes above 2^47 (the reference's {60, b ..., 60})
            bits = [int(rng.choice([36, 40, 44, 45, 46, 60], p=[0.15, 0.2, 0.1, 0.2, 0.1, 0.25])) for _ in range(n_data)]
        bits.append(int(rng.choice([45, 47, 50, 58, 60])))  # the special prime
        force = bool(rng.random() < 0.15)
        if force:
            os.environ["HE355_FORCE_U64"] = "1"
        try:
            try:
                g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
            except be.HE355Error as e:  # e.g. not enough primes of a size for this N
                print(f"case {case}: N={N} bits={bits} skipped ({str(e)[:60]})", flush=True)
                continue
        finally:
            os.environ.pop("HE355_FORCE_U64", None)
        o = oracle.Context(oracle.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
        assert g.moduli == o.moduli
        L = int(rng.integers(1, g.L + 1))
        n = int(rng.choice([1, 2, 3, 5, 8, 9, 13]))
        g.set_chunk(int(rng.choice([2, 4, 32, 256])))
        g.set_latency_max(int(rng.choice([0, 4, 8])))
        rk = o.random_kswitch_key(rng)
        g.set_relin_key(rk)
        a = np.stack([o.random_poly(rng, L, 2) for _ in range(n)])
        b = np.stack([o.random_poly(rng, L, 2) for _ in range(n)])
        da, db = g.to_device(a), g.to_device(b)
        what = []
        # multiply -> relinearize (-> rescale): pairwise, or HEBench's outer product (result r = i * b1 + x) with operand bases > 0
        b1 = int(rng.choice([0, 0, 1, 2, 3]))  # 0: pairwise
        if b1 and n >= b1 + 2:
            a_base, b_base = 1, int(rng.integers(0, n - b1 + 1))
            nres = ((n - 1) // b1) * b1 if (n - 1) // b1 else b1
            nres = min(nres, (n - 1) * b1)
            ix = be.Context.outer(a_base, nres // b1, b_base, b1)
            pick = [(a_base + r // b1, b_base + r % b1) for r in range(nres)]
            what.append(f"outer{b1}")
        else:
            nres, ix = n, be.Context.pairwise()
            pick = [(r, r) for r in range(n)]
        out = g.alloc(nres * 2 * L * N)
        g.multiply_relin(L, nres, da, db, ix, out)
        got = out.download((nres, 2, L, N))
        want = [o.relinearize(o.multiply_ntt(a[i], b[x]), rk) for i, x in pick]
        ok = all(np.array_equal(got[r], want[r]) for r in range(nres))
        what.append("mul_relin")
        if ok and L >= 2:
            out2 = g.alloc(nres * 2 * (L - 1) * N)
            g.multiply_relin(L, nres, da, db, ix, out2, rescale=True)
            got2 = out2.download((nres, 2, L - 1, N))
            ok = all(np.array_equal(got2[r], o.rescale(want[r])) for r in range(nres))
            what.append("rescale")
        # relinearize of size-3 ciphertexts
        if ok:
            m = min(n, 3)
            ct3 = np.stack([o.random_poly(rng, L, 3) for _ in range(m)])
            o3 = g.alloc(m * 2 * L * N)
            g.relinearize(L, m, g.to_device(ct3), o3)
            g3 = o3.download((m, 2, L, N))
            ok = all(np.array_equal(g3[r], o.relinearize(ct3[r], rk)) for r in range(m))
            what.append("relin3")
        # a rotation with its own key, and rotate_add
        if ok:
            step = int(rng.choice([1, -1, 2, 4, -8, 16]))
            e = o.galois_elt(step)
            gk = o.random_kswitch_key(rng)
            g.set_galois_key(e, gk)
            orot = g.alloc(n * 2 * L * N)
            g.rotate(L, n, da, step, orot)
            gr = orot.download((n, 2, L, N))
            wr = [o.apply_galois(a[r], e, gk) for r in range(n)]
            ok = all(np.array_equal(gr[r], wr[r]) for r in range(n))
            what.append(f"rot{step}")
            if ok:
                g.rotate_add(L, n, da, step, db, orot)
                gr = orot.download((n, 2, L, N))
                ok = all(np.array_equal(gr[r], o.add(b[r], wr[r])) for r in range(n))
                what.append("rot_add")
        if ok and rng.random() < 0.6:
            ok, extra = chains(be, g, o, rng, L, n, a, da)
            what += extra
        print(f"case {case}: N={N} bits={bits} force_u64={int(force)} L={L} n={n} {'+'.join(what)} {'ok' if ok else 'MISMATCH'}", flush=True)
        g.close()
        if not ok:
            return 1
    print(f"{case} cases, all equal")
    return 0


if __name__ == "__main__":
    sys.exit(main())
