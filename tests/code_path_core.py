"""Compact core of the GPU parity suite, run as a CHILD PROCESS by tests/test_gpu_code_paths.py once per kernel-selecting setting
(the HE355_* switches are read when a context is created, some once per process): NTT round trip, multiply -> relinearize ->
rescale (n = 5), a rotation that takes the NAF path, BFV multiply + relinearize, one he355_rotate_sum level walk and one DotProduct
through the API-Bridge C ABI -- each compared bit for bit with the oracle (the bridge result with cleartext).  Exit code 0 = all equal.
usage: python tests/code_path_core.py            (environment = the setting under test)"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def rand_cts(o, rng, n, L, size=2):
    return np.stack([o.random_poly(rng, L, size) for _ in range(n)])


def default_galois_keys(g, o, rng):
    keys, k = {}, 0
    while (1 << k) < g.N // 2:
        for s in (1 << k, -(1 << k)):
            e = o.galois_elt(s)
            keys[e] = o.random_kswitch_key(rng)
            g.set_galois_key(e, keys[e])
        k += 1
    return keys


def main():
    import oracle as ho
    be = importlib.import_module("reference-seal-backend_amd")
    if be.device_count() < 1:
        raise SystemExit("no HIP device")
    rng = np.random.default_rng(20251005)
    done = []
    # ---- CKKS, N = 4096, {60, 45, 45, 60}: both arithmetic engines in one chain ----
    N, bits = 4096, [60, 45, 45, 60]
    g = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, sec128=False, device=0)
    o = ho.Context(ho.SCHEME_CKKS, N, bit_sizes=bits, sec128=False)
    assert g.moduli == o.moduli
    L, K = g.L, g.K
    polys = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in g.moduli])[None]
    d = g.to_device(polys)
    g.ntt(d, K, list(range(K)))
    f = d.download(polys.shape)
    for i in range(K):
        assert np.array_equal(f[0, i], o.ntt(i, polys[0, i])), ("ntt", i)
    g.ntt(d, K, list(range(K)), inverse=True)
    assert np.array_equal(d.download(polys.shape), polys)
    done.append("ntt")
    rk = o.random_kswitch_key(rng)
    g.set_relin_key(rk)
    n = 5
    a, b = rand_cts(o, rng, n, L), rand_cts(o, rng, n, L)
    out = g.alloc(n * 2 * (L - 1) * N)
    g.multiply_relin(L, n, g.to_device(a), g.to_device(b), be.Context.pairwise(0, 0), out, rescale=True)
    got = out.download((n, 2, L - 1, N))
    for r in range(n):
        assert np.array_equal(got[r], o.rescale(o.relinearize(o.multiply_ntt(a[r], b[r]), rk))), ("mul_relin_rescale", r)
    done.append("mul_relin_rescale")
    keys = default_galois_keys(g, o, rng)
    rot = g.alloc(n * 2 * L * N)
    g.rotate(L, n, g.to_device(a), 3, rot)  # 3 = 4 - 1: two NAF terms, no key of its own
    got = rot.download((n, 2, L, N))
    for r in range(n):
        assert np.array_equal(got[r], o.rotate(a[r], 3, keys)), ("rotate_naf", r)
    done.append("rotate_naf")
    g.close()
    # ---- BFV, N = 4096, {60, 40, 60}: BEHZ multiply of a 2 x 2 outer product + relinearize, one level walk ----
    bb = [60, 40, 40, 60]
    gb = be.Context(be.SCHEME_BFV, N, bit_sizes=bb, plain_bits=20, sec128=False, device=0)
    ob = ho.Context(ho.SCHEME_BFV, N, bit_sizes=bb, plain_bits=20, sec128=False)
    Lb = gb.L
    xa, xb = rand_cts(ob, rng, 2, Lb), rand_cts(ob, rng, 2, Lb)
    rkb = ob.random_kswitch_key(rng)
    gb.set_relin_key(rkb)
    c3, c2 = gb.alloc(4 * 3 * Lb * N), gb.alloc(4 * 2 * Lb * N)
    gb.bfv_multiply(Lb, 4, gb.to_device(xa), gb.to_device(xb), be.Context.outer(0, 2, 0, 2), c3)
    gb.relinearize(Lb, 4, c3, c2)
    got = c2.download((4, 2, Lb, N))
    for r in range(4):
        assert np.array_equal(got[r], ob.relinearize(ob.bfv_multiply(xa[r // 2], xb[r % 2]), rkb)), ("bfv_multiply_relin", r)
    done.append("bfv_multiply_relin")
    keysb = default_galois_keys(gb, ob, rng)
    gb.set_latency_max(0)
    nct = 10
    src = rand_cts(ob, rng, nct, Lb)
    acc = gb.alloc(nct * 2 * Lb * N)
    spacers = (N // 2) // 8
    steps = [j * spacers for j in range(1, 8)]
    gb.rotate_sum(Lb, nct, gb.to_device(src), steps, acc)
    got = acc.download((nct, 2, Lb, N))
    for r in (0, nct - 1):
        want = src[r].copy()
        for s in steps:
            want = ob.add(want, ob.rotate(src[r], s, keysb))
        assert np.array_equal(got[r], want), ("rotate_sum", r)
    done.append("rotate_sum")
    gb.close()
    # ---- one descriptor through the API-Bridge C ABI: CKKS DotProduct offline 2 x 2 (multiply, relinearize, accumulate) ----
    from hebench_harness import Backend, OFFLINE, SCHEME_CKKS, W_DOT
    bk = Backend(be.LIB_PATH)
    nvec = 100
    va, vb = rng.uniform(-1, 1, (2, nvec)), rng.uniform(-1, 1, (2, nvec))
    hb = bk.create(bk.find(W_DOT, SCHEME_CKKS, OFFLINE), [("n", nvec), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 40),
                                                          ("ScaleBits", 40), ("NumThreads", 0)], (2, 2))
    res = bk.run(hb, [va, vb], 1, np.float64)
    assert np.allclose(res, (va @ vb.T).reshape(4, 1), atol=1e-3), "bridge_dot"  # (CKKS is approximate by construction: tolerance of the bridge tests)
    bk.destroy(hb)
    bk.close()
    done.append("bridge_dot")
    print("code paths ok:", " ".join(done))


if __name__ == "__main__":
    main()
