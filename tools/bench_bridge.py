#!/usr/bin/env python3
"""What a HEBench test_harness times, for every descriptor the engine registers (csrc/bridge/engine.cpp, same 20 as
/root/reference/src/engine/seal_engine.cpp:108-151): operate() through the API-Bridge C ABI, replayed with the harness's protocol --

  Latency : batch 1 per operand, 1 warm-up iteration (warmup_iterations_count = 1, ckks eltwise .cpp:40) then REPS single operate()
            calls, each result handle destroyed before the next call; median and minimum wall time per call;
  Offline : one operate() over the whole batch (ckks eltwise .cpp:322-336), 1 warm-up + REPS calls, median / minimum;

-- next to (a) the phases around it (encode, encrypt, load, store, decrypt, decode; wall clock, once), (b) the raw hipMalloc / hipFree
calls the timed operate() calls made (he355_alloc_stats, process totals: 0 in steady state with the device pool, csrc/device_pool.h),
and (c) `direct_ms`: the same evaluator sequence issued straight at the he355_* C ABI on pre-allocated slabs with synthetic operands of
the same shape (what bench.py and tools/latency_probe.py time), where operate() is one or two he355 calls.  Results are checked against
the cleartext ground truth the way the harness does.

Usage (GPU box):  python tools/bench_bridge.py [--sizes default|bench|both] [--only SUBSTR] [--reps 20] [--out FILE]
                  HE355_POOL=0 python tools/bench_bridge.py ...      # the pre-pool behaviour (hipMalloc / drain + hipFree per call)
One JSON object per line."""
import argparse
import ctypes as C
import importlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hebench_harness import (Backend, Handle, ParameterIndexer, LATENCY, OFFLINE, SCHEME_BFV, SCHEME_CKKS, W_ADD, W_DOT, W_MATMUL, W_MUL,  # noqa: E402
                             W_LOGREG3)

be = importlib.import_module("reference-seal-backend_amd")

W_NAMES = {W_ADD: "EltwiseAdd", W_MUL: "EltwiseMultiply", W_DOT: "DotProduct", W_MATMUL: "MatrixMultiply", W_LOGREG3: "LogisticRegression_PolyD3"}
MM_NAMES = {0: "MatMultVal", 1: "CipherBatchAxis", 2: "MatMultRow"}


def centre(v, t):
    v = np.mod(v, t)
    return np.where(v > t // 2, v - t, v)


def plain_t(N):
    return 1032193 if N <= 8192 else 786433


class Case:
    """one descriptor + one parameter set: operands, sample counts, ground truth"""

    def __init__(self, bench, params, label):
        self.bench, self.params, self.label = bench, list(params), label
        d = bench["desc"]
        self.workload, self.scheme, self.category, self.other = d.workload, d.scheme, d.category, d.other
        self.p = dict(params)
        rng = np.random.default_rng(1234)  # the reference's CI seed (.github/workflows/cmake.yml:43)
        ckks = self.scheme == SCHEME_CKKS
        lat = self.category == LATENCY
        N = self.p["PolyModulusDegree"]
        self.tol = 1e-3
        if self.workload in (W_ADD, W_MUL, W_DOT):
            n = self.p["n"]
            counts = (1, 1) if lat else self.p.pop("_counts")
            if ckks:
                a, b = rng.uniform(-1, 1, (counts[0], n)), rng.uniform(-1, 1, (counts[1], n))
            else:
                lim = 20 if self.workload == W_DOT else 700
                a, b = rng.integers(-lim, lim, (counts[0], n)).astype(np.int64), rng.integers(-lim, lim, (counts[1], n)).astype(np.int64)
            self.operands, self.counts = [a, b], counts
            if self.workload == W_ADD:
                want, self.out_n = (a[:, None, :] + b[None, :, :]).reshape(-1, n), n
            elif self.workload == W_MUL:
                want, self.out_n = (a[:, None, :] * b[None, :, :]).reshape(-1, n), n
            else:
                want, self.out_n = (a @ b.T).reshape(-1, 1), 1
                self.tol = 1e-2
            self.want = want if ckks else centre(want, plain_t(N))
        elif self.workload == W_MATMUL:
            r0, c0, c1 = self.p["rows_M0"], self.p["cols_M0"], self.p["cols_M1"]
            if ckks:
                A, B = rng.uniform(-1, 1, (r0, c0)), rng.uniform(-1, 1, (c0, c1))
                self.want = (A @ B).reshape(1, -1)
            else:
                A, B = rng.integers(-8, 8, (r0, c0)).astype(np.int64), rng.integers(-8, 8, (c0, c1)).astype(np.int64)
                self.want = centre(A @ B, plain_t(N)).reshape(1, -1)
            self.operands, self.counts, self.out_n = [A.reshape(1, -1), B.reshape(1, -1)], (1, 1), r0 * c1
        else:  # LogisticRegression_PolyD3
            n = self.p["n"]
            batch = 1 if lat else self.p.pop("_counts")[2]
            W, bias, X = rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (1, 1)), rng.uniform(-1, 1, (batch, n))
            x = X @ W[0] + bias[0, 0]
            self.want = (0.5 + 0.15012 * x - 0.0015930078125 * x ** 3).reshape(-1, 1)  # SigmoidPolyCoeff, logreg .h:117
            self.operands, self.counts, self.out_n = [W, bias, X], (1, 1, batch), 1
        self.params = [(k, v) for k, v in self.params if not k.startswith("_")]
        self.dtype = np.float64 if ckks else np.int64

    def name(self):
        s = "CKKS" if self.scheme == SCHEME_CKKS else "BFV"
        w = W_NAMES[self.workload] + (f"/{MM_NAMES[self.other]}" if self.workload == W_MATMUL else "")
        return f"{s} {w} {'Latency' if self.category == LATENCY else 'Offline'}"

    def correct(self, res):
        if self.scheme == SCHEME_CKKS:
            return bool(np.allclose(res.reshape(self.want.shape), self.want, atol=self.tol))
        return bool(np.array_equal(res.reshape(self.want.shape), self.want))


def cases(backend, sizes):
    """the parameter sets: every descriptor at the reference's registered defaults; `bench`: the BASELINE.json configurations the
    descriptor maps to"""
    out = []
    for b in backend.benchmarks():
        d = b["desc"]
        dflt = [(k.replace("CoefficientMudulusBits", "CoefficientModulusBits"), v) for k, v in b["defaults"][0]]
        vec = d.workload in (W_ADD, W_MUL, W_DOT)
        if sizes in ("default", "both"):
            extra = []
            if d.category == OFFLINE:
                extra = [("_counts", (1, 1, 100) if d.workload == W_LOGREG3 else (8, 8) if d.workload == W_DOT else (16, 16))]
            out.append(Case(b, dflt + extra, "reference defaults"))
        if sizes in ("bench", "both"):
            ck = d.scheme == SCHEME_CKKS
            p = dict(dflt)
            if vec and ck and d.workload == W_MUL:  # configs[1]: CKKS EltwiseMult, N=2^14, L=8, 256 results
                p.update(n=8192, PolyModulusDegree=16384, MultiplicativeDepth=8, CoefficientModulusBits=45, ScaleBits=45)
                cnt, lab = (16, 16), "BASELINE configs[1] (N=2^14, L=8, 16x16 = 256 results)"
            elif vec and ck and d.workload == W_DOT:  # configs[3]
                p.update(n=4096, PolyModulusDegree=32768, MultiplicativeDepth=16, CoefficientModulusBits=45, ScaleBits=45)
                cnt, lab = (8, 8), "BASELINE configs[3] shape (n=4096, N=2^15, L=16, 8x8 = 64 results)"
            elif vec and ck and d.workload == W_ADD:
                p.update(n=8192, PolyModulusDegree=16384, MultiplicativeDepth=8, CoefficientModulusBits=45, ScaleBits=45)
                cnt, lab = (16, 16), "N=2^14, L=8, 16x16"
            elif vec and not ck and d.workload == W_ADD:  # configs[0] is the default-parameter BFV add
                continue
            elif vec and not ck:
                p.update(n=4096, PolyModulusDegree=16384, MultiplicativeDepth=4, CoefficientModulusBits=40)
                cnt, lab = ((8, 8), "N=2^14, depth 4, 8x8")
            elif d.workload == W_MATMUL and d.other == 2 and not ck:  # configs[4]
                p.update(rows_M0=128, cols_M0=128, cols_M1=128, PolyModulusDegree=32768, MultiplicativeDepth=3, CoefficientModulusBits=40)
                cnt, lab = None, "BASELINE configs[4] (128x128x128, N=2^15, depth 3)"
            elif d.workload == W_MATMUL and d.other == 2 and ck:
                p.update(rows_M0=64, cols_M0=64, cols_M1=64, PolyModulusDegree=16384, MultiplicativeDepth=3)
                cnt, lab = None, "64x64x64, N=2^14, depth 3"
            elif d.workload == W_MATMUL and d.other == 0:
                p.update(rows_M0=32, cols_M0=512, cols_M1=32, PolyModulusDegree=16384, MultiplicativeDepth=3)
                cnt, lab = None, "32x512x32, N=2^14, depth 3"
            elif d.workload == W_MATMUL and d.other == 1:
                p.update(rows_M0=16, cols_M0=16, cols_M1=16)
                cnt, lab = None, "16x16x16 at the default parameters"
            else:
                continue
            pl = [(k, p[k]) for k, _ in dflt]
            if d.category == OFFLINE and cnt:
                pl.append(("_counts", cnt))
            out.append(Case(b, pl, lab))
    return out


def run_case(backend, case, reps):
    L = backend.L
    t = {}

    def clock(name, fn):
        t0 = time.perf_counter()
        backend.chk(fn())
        t[name] = (time.perf_counter() - t0) * 1e3

    hb = backend.create(case.bench, case.params, case.counts)
    dpc, keep = backend.pack(case.operands)
    h_plain, h_cipher, h_remote, h_out = Handle(), Handle(), Handle(), Handle()
    clock("encode_ms", lambda: L.encode(hb, C.byref(dpc), C.byref(h_plain)))
    clock("encrypt_ms", lambda: L.encrypt(hb, h_plain, C.byref(h_cipher)))
    clock("load_ms", lambda: L.load(hb, C.byref(h_cipher), 1, C.byref(h_remote)))
    idx = [(0, o.shape[0]) for o in case.operands]
    pi = (ParameterIndexer * len(idx))(*[ParameterIndexer(v, b) for v, b in idx])
    times = []
    st0 = st1 = None
    for r in range(reps + 2):  # call 0: cold (arenas, keys' scaled copies ...); call 1: the harness's warm-up iteration; then the timed calls
        if h_out.p:
            L.destroyHandle(h_out)
            h_out = Handle()
        if r == 2:
            st0 = be.process_alloc_stats()
        t0 = time.perf_counter()
        backend.chk(L.operate(hb, h_remote, pi, len(idx), C.byref(h_out)))
        dt = (time.perf_counter() - t0) * 1e3
        if r == 0:
            t["operate_cold_ms"] = dt
        elif r >= 2:
            times.append(dt)
    st1 = be.process_alloc_stats()
    t["operate_ms"] = statistics.median(times)
    t["operate_min_ms"] = min(times)
    local = (Handle * 1)()
    clock("store_ms", lambda: L.store(hb, h_out, local, 1))
    h_dec = Handle()
    clock("decrypt_ms", lambda: L.decrypt(hb, local[0], C.byref(h_dec)))
    n_res = case.want.shape[0] if case.workload != W_MATMUL else 1
    res = np.zeros((n_res, case.out_n), dtype=case.dtype)
    res.fill(0)  # (a harness hands over buffers it has written: np.zeros alone leaves the pages unmapped and decode() would pay the faults)
    out_pack, keep2 = backend.pack([res])
    clock("decode_ms", lambda: L.decode(hb, h_dec, C.byref(out_pack)))
    clock("decode_again_ms", lambda: L.decode(hb, h_dec, C.byref(out_pack)))  # the same call once more: what a second decode of the run costs
    for h in (h_plain, h_cipher, h_remote, h_out, local[0], h_dec):
        L.destroyHandle(h)
    backend.destroy(hb)
    return dict(results=int(n_res), correct=case.correct(res), reps=reps,
                raw_mallocs_in_timed_calls=st1["raw_mallocs"] - st0["raw_mallocs"], raw_frees_in_timed_calls=st1["raw_frees"] - st0["raw_frees"],
                pool_hits_in_timed_calls=st1["pool_hits"] - st0["pool_hits"], **t)


def direct(case, reps):
    """the same evaluator sequence straight at the he355 C ABI (pre-allocated slabs, synthetic operands and keys of the same shape):
    median wall time of call + stream sync, and the HIP-event time of the same region"""
    p = case.p
    ckks = case.scheme == SCHEME_CKKS
    if case.workload == W_MATMUL and not (case.other == 0 and ckks):
        return None
    if case.workload == W_LOGREG3:
        return None
    N, depth, bits = p["PolyModulusDegree"], p["MultiplicativeDepth"], p["CoefficientModulusBits"]
    chain = [60] + [bits] * (depth - 1) + [60]
    g = be.Context(be.SCHEME_CKKS if ckks else be.SCHEME_BFV, N, bit_sizes=chain, plain_bits=0 if ckks else 20, device=0)
    try:
        L = g.L
        if case.workload == W_MATMUL:
            b0, b1, count = p["rows_M0"], p["cols_M1"], p["cols_M0"]
        else:
            b0, b1, count = case.counts[0], case.counts[1], p["n"]
        n = b0 * b1
        per = 2 * L * N
        a, b = g.alloc(b0 * per), g.alloc(b1 * per)
        pm = list(range(L))
        g.fill_uniform(a, b0 * 2 * L, pm, 1)
        g.fill_uniform(b, b1 * 2 * L, pm, 2)
        ix = be.Context.outer(0, b0, 0, b1)
        need_keys = case.workload in (W_DOT, W_MATMUL)
        if need_keys:
            g.set_relin_key_synthetic(7)
            row = N // 2  # CKKS: slots; BFV: one batching row (seal_context.cpp:295, 324)
            cnt = min(count, row)
            rot = cnt.bit_length() - (1 if cnt & (cnt - 1) == 0 else 0)
            for i in range(rot):
                g.set_galois_key_synthetic(g.galois_elt(1 << i), 100 + i)
            if not ckks and count > row:
                g.set_galois_key_synthetic(2 * N - 1, 99)
        rescale = case.workload == W_MATMUL
        Lo = L - 1 if rescale else L
        out = g.alloc(n * (3 if case.workload == W_MUL else 2) * Lo * N)
        tmp = g.alloc(n * 2 * Lo * N) if need_keys else None
        c3 = g.alloc(n * 3 * L * N) if (need_keys and not ckks) else None

        def once():
            if case.workload == W_ADD:
                g.add(L, 2, n, a, b, ix, out)
            elif case.workload == W_MUL:
                (g.multiply if ckks else g.bfv_multiply)(L, n, a, b, ix, out)
            else:
                if ckks:
                    g.multiply_relin(L, n, a, b, ix, out, rescale=rescale)
                else:
                    g.bfv_multiply(L, n, a, b, ix, c3)
                    g.relinearize(L, n, c3, out)
                g.accumulate(Lo, n, out, count, tmp)
            g.sync()

        once()
        once()
        wall, ev = [], []
        for _ in range(reps):
            g.timer_begin()
            t0 = time.perf_counter()
            once()
            wall.append((time.perf_counter() - t0) * 1e3)
            ev.append(g.timer_end())
        return dict(direct_ms=statistics.median(wall), direct_min_ms=min(wall), direct_event_ms=statistics.median(ev))
    finally:
        g.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="default", choices=["default", "bench", "both"])
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-direct", action="store_true")
    ap.add_argument("--exact-reps", action="store_true", help="--reps timed calls for Offline descriptors too (tools/bridge_ktrace.sh)")
    a = ap.parse_args()
    if be.device_count() < 1:
        raise SystemExit("bench_bridge.py needs an MI355X (no CPU fallback)")
    os.environ.setdefault("HE355_SEED", "1234")
    backend = Backend(be.LIB_PATH)
    pool = "off (HE355_POOL=0)" if os.environ.get("HE355_POOL", "1")[:1] == "0" else "on"
    fout = open(a.out, "a") if a.out else None
    for i, case in enumerate(cases(backend, a.sizes)):
        if a.only and a.only.lower() not in (case.name() + " " + case.label).lower():
            continue
        reps = a.reps if (case.category == LATENCY or a.exact_reps) else max(3, a.reps // 4)
        r = run_case(backend, case, reps)
        rec = dict(descriptor=case.name(), sizes=case.label, params=dict(case.params), sample_counts=list(case.counts), pool=pool, **r)
        if not a.no_direct:
            d = direct(case, reps)
            rec.update(d if d else dict(direct_ms=None))
            if d:
                rec["operate_over_direct"] = rec["operate_ms"] / d["direct_ms"]
        line = json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items()})
        print(line, flush=True)
        if fout:
            fout.write(line + "\n")
            fout.flush()
    backend.close()


if __name__ == "__main__":
    main()
