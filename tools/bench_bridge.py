#!/usr/bin/env python3
"""Phase timings through the HEBench API-Bridge C ABI (the drop-in boundary): the call sequence test_harness performs
(encode, encrypt, load, operate, store, decrypt, decode), wall clock per phase, operate() repeated like
the harness's measurement loop.  One JSON object per workload; the output goes to profiles/ as evidence for DESIGN.md.
Usage (GPU box): python tools/bench_bridge.py [name ...]"""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hebench_harness import (Backend, Handle, ParameterIndexer, LATENCY, OFFLINE, SCHEME_BFV, SCHEME_CKKS, W_ADD, W_DOT, W_MATMUL, W_MUL,  # noqa: E402
                             W_LOGREG3)

be = importlib.import_module("reference-seal-backend_amd")


def phases(backend, hb, operands, out_n, out_dtype, reps=3):
    L = backend.L
    t = {}

    def clock(name, fn):
        t0 = time.perf_counter()
        backend.chk(fn())
        t[name] = (time.perf_counter() - t0) * 1e3

    dpc, keep = backend.pack(operands)
    h_plain, h_cipher, h_remote, h_out = Handle(), Handle(), Handle(), Handle()
    clock("encode_ms", lambda: L.encode(hb, C.byref(dpc), C.byref(h_plain)))
    clock("encrypt_ms", lambda: L.encrypt(hb, h_plain, C.byref(h_cipher)))
    clock("load_ms", lambda: L.load(hb, C.byref(h_cipher), 1, C.byref(h_remote)))
    idx = [(0, o.shape[0]) for o in operands]
    pi = (ParameterIndexer * len(idx))(*[ParameterIndexer(v, b) for v, b in idx])
    best = None
    for r in range(reps + 1):  # first call untimed (warm-up), as the harness's warm-up iteration
        if h_out.p:
            L.destroyHandle(h_out)
            h_out = Handle()
        t0 = time.perf_counter()
        backend.chk(L.operate(hb, h_remote, pi, len(idx), C.byref(h_out)))
        dt = (time.perf_counter() - t0) * 1e3
        if r and (best is None or dt < best):
            best = dt
    t["operate_ms"] = best
    local = (Handle * 1)()
    clock("store_ms", lambda: L.store(hb, h_out, local, 1))
    h_dec = Handle()
    clock("decrypt_ms", lambda: L.decrypt(hb, local[0], C.byref(h_dec)))
    n_res = int(np.prod([b for _, b in idx]))
    res = np.zeros((n_res, out_n), dtype=out_dtype)
    out_pack, keep2 = backend.pack([res])
    clock("decode_ms", lambda: L.decode(hb, h_dec, C.byref(out_pack)))
    for h in (h_plain, h_cipher, h_remote, h_out, local[0], h_dec):
        L.destroyHandle(h)
    return res, t, n_res


def ckks(n, N, depth, bits):
    return [("n", n), ("PolyModulusDegree", N), ("MultiplicativeDepth", depth), ("CoefficientModulusBits", bits), ("ScaleBits", bits), ("NumThreads", 0)]


def w_add(b):
    rng = np.random.default_rng(1)
    n, cnt = 4096, (64, 64)
    a, c = rng.uniform(-1, 1, (cnt[0], n)), rng.uniform(-1, 1, (cnt[1], n))
    hb = b.create(b.find(W_ADD, SCHEME_CKKS, OFFLINE), ckks(n, 8192, 2, 45), cnt)
    res, t, nres = phases(b, hb, [a, c], n, np.float64)
    ok = bool(np.allclose(res, (a[:, None, :] + c[None, :, :]).reshape(-1, n), atol=1e-4))
    b.destroy(hb)
    return dict(workload="CKKS EltwiseAdd offline 64x64, n=4096, N=8192, depth 2", results=nres, correct=ok, **t)


def w_mul(b):
    rng = np.random.default_rng(2)
    n, cnt = 8192, (16, 16)
    a, c = rng.uniform(-1, 1, (cnt[0], n)), rng.uniform(-1, 1, (cnt[1], n))
    hb = b.create(b.find(W_MUL, SCHEME_CKKS, OFFLINE), ckks(n, 16384, 7, 45), cnt)
    res, t, nres = phases(b, hb, [a, c], n, np.float64)
    ok = bool(np.allclose(res, (a[:, None, :] * c[None, :, :]).reshape(-1, n), atol=1e-4))
    b.destroy(hb)
    return dict(workload="CKKS EltwiseMult offline 16x16, n=8192, N=2^14, L=8 (BASELINE configs[1])", results=nres, correct=ok, **t)


def w_dot(b):
    rng = np.random.default_rng(3)
    n, cnt = 4096, (8, 8)
    a, c = rng.uniform(-1, 1, (cnt[0], n)), rng.uniform(-1, 1, (cnt[1], n))
    hb = b.create(b.find(W_DOT, SCHEME_CKKS, OFFLINE), ckks(n, 32768, 15, 45), cnt)
    res, t, nres = phases(b, hb, [a, c], 1, np.float64)
    ok = bool(np.allclose(res, (a @ c.T).reshape(-1, 1), atol=1e-2))
    b.destroy(hb)
    return dict(workload="CKKS DotProduct offline 8x8, n=4096, N=2^15, L=16 (BASELINE configs[3])", results=nres, correct=ok, **t)


def w_logreg(b):
    rng = np.random.default_rng(4)
    n, batch = 16, 100
    W, bias, X = rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (1, 1)), rng.uniform(-1, 1, (batch, n))
    hb = b.create(b.find(W_LOGREG3, SCHEME_CKKS, OFFLINE), ckks(n, 16384, 6, 45), (1, 1, batch))
    res, t, nres = phases(b, hb, [W, bias, X], 1, np.float64)
    x = X @ W[0] + bias[0, 0]
    want = 0.5 + 0.15012 * x - 0.0015930078125 * x ** 3  # sigmoid polynomial of degree 3 (SigmoidPolyCoeff, logreg .h:117)
    ok = bool(np.allclose(res[:, 0], want, atol=1e-3))
    b.destroy(hb)
    return dict(workload="CKKS LogisticRegression_PolyD3 offline, 16 features, 100 samples, N=2^14, depth 6 (reference defaults)", results=nres,
                correct=ok, **t)


def w_matmul(b):
    rng = np.random.default_rng(5)
    r0 = c0 = c1 = 128
    A, B = rng.integers(-8, 8, (r0, c0)).astype(np.int64), rng.integers(-8, 8, (c0, c1)).astype(np.int64)
    bench = [x for x in b.benchmarks() if x["desc"].workload == W_MATMUL and x["desc"].other == 2 and x["desc"].scheme == SCHEME_BFV][0]
    hb = b.create(bench, [("rows_M0", r0), ("cols_M0", c0), ("cols_M1", c1), ("PolyModulusDegree", 32768), ("MultiplicativeDepth", 3),
                          ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0)])
    res, t, nres = phases(b, hb, [A.reshape(1, -1), B.reshape(1, -1)], r0 * c1, np.int64, reps=2)
    tm = 786433
    want = np.mod(A @ B, tm)
    want = np.where(want > tm // 2, want - tm, want)
    ok = bool(np.array_equal(res.reshape(r0, c1), want))
    b.destroy(hb)
    return dict(workload="BFV MatMult (row-major) latency 128x128x128, N=2^15, depth 3 (BASELINE configs[4])", results=nres, correct=ok, **t)


if __name__ == "__main__":
    if be.device_count() < 1:
        raise SystemExit("bench_bridge.py needs an MI355X (no CPU fallback)")
    groups = {"add": w_add, "mul": w_mul, "dot": w_dot, "logreg": w_logreg, "matmul": w_matmul}
    backend = Backend(be.LIB_PATH)
    for name in (sys.argv[1:] or list(groups)):
        r = groups[name](backend)
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
    backend.close()
