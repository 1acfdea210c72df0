#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations (configs[0], [1], [3], [4]) on one MI355X, inputs resident in HBM,
HIP events on the library's stream.  Not the judged bench line (that is bench.py = configs[2]); the output goes to
profiles/ as evidence for DESIGN.md.  One JSON object per line."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
be = importlib.import_module("reference-seal-backend_amd")
HBM_PEAK = 8.0e12


def timed(ctx, fn, reps):
    fn()
    ctx.sync()
    ctx.timer_begin()
    for _ in range(reps):
        fn()
    ms = ctx.timer_end()
    return ms / reps


def cfg1_bfv_add():
    # configs[0] at the reference's default parameters (N=8192, {60,40,60}), batch scaled up to 4096 results
    g = be.Context(be.SCHEME_BFV, 8192, bit_sizes=[60, 40, 60], plain_bits=20, device=0)
    L, N, n = g.L, g.N, 4096
    a, b, o = g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
    g.fill_uniform(a, n * 2 * L, list(range(L)), 1)
    g.fill_uniform(b, n * 2 * L, list(range(L)), 2)
    ms = timed(g, lambda: g.add(L, 2, n, a, b, be.Context.pairwise(), o), 20)
    byts = 3 * 2 * L * N * 8
    out = dict(config="configs[0] BFV EltwiseAdd N=8192 L=2", results=n, ms=ms, ops_per_s=n / ms * 1e3, algorithmic_bytes_per_op=byts,
               hbm_GBps=byts * n / ms / 1e6, roofline_frac=byts * n / (ms / 1e3) / HBM_PEAK)
    g.close()
    return out


def cfg2_ckks_multiply():
    g = be.Context(be.SCHEME_CKKS, 16384, bit_sizes=be.chain_bits(8, 45), device=0)
    L, N = g.L, g.N
    res = []
    for b0, b1 in ((256, 1), (16, 16)):
        n = b0 * b1
        a, b, o = g.alloc(b0 * 2 * L * N), g.alloc(b1 * 2 * L * N), g.alloc(n * 3 * L * N)
        g.fill_uniform(a, b0 * 2 * L, list(range(L)), 1)
        g.fill_uniform(b, b1 * 2 * L, list(range(L)), 2)
        ms = timed(g, lambda: g.multiply(L, n, a, b, be.Context.outer(0, b0, 0, b1), o), 20)
        byts = 7 * L * N * 8
        res.append(dict(config=f"configs[1] CKKS EltwiseMult N=2^14 L=8 batch {b0}x{b1}", results=n, ms=ms, ops_per_s=n / ms * 1e3,
                        algorithmic_bytes_per_op=byts, hbm_GBps=byts * n / ms / 1e6, roofline_frac=byts * n / (ms / 1e3) / HBM_PEAK))
    g.close()
    return res


def cfg4_dot_product():
    g = be.Context(be.SCHEME_CKKS, 32768, bit_sizes=be.chain_bits(16, 45), device=0)
    L, N, n, count = g.L, g.N, 64, 4096
    a, b, o, t = g.alloc(n * 2 * L * N), g.alloc(2 * L * N), g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
    g.fill_uniform(a, n * 2 * L, list(range(L)), 1)
    g.fill_uniform(b, 2 * L, list(range(L)), 2)
    g.set_relin_key_synthetic(7)
    for k in range(12):
        g.set_galois_key_synthetic(g.galois_elt(1 << k), 100 + k)

    def run():
        g.multiply_relin(L, n, a, b, be.Context.outer(0, n, 0, 1), o)
        g.accumulate(L, n, o, count, t)
    ms = timed(g, run, 3)
    out = dict(config="configs[3] CKKS DotProduct n=4096 N=2^15 L=16 (multiply, relinearize, 12 rotate+add)", results=n, ms=ms, ops_per_s=n / ms * 1e3,
               key_switches_per_result=13)
    g.close()
    return out


def headline_latency():
    """HEBench's Latency category on the headline operation (configs[2] with batch 1): one multiply + relinearize + rescale per
    call, host-synchronised after every call (wall clock), and the same for small offline batches."""
    import time
    g = be.Context(be.SCHEME_CKKS, 32768, bit_sizes=be.chain_bits(16, 45), device=0)
    L, N = g.L, g.N
    g.set_relin_key_synthetic(7)
    res = []
    for n in (1, 8, 64):
        a, b, o = g.alloc(n * 2 * L * N), g.alloc(2 * L * N), g.alloc(n * 2 * (L - 1) * N)
        g.fill_uniform(a, n * 2 * L, list(range(L)), 1)
        g.fill_uniform(b, 2 * L, list(range(L)), 2)
        ix = be.Context.outer(0, n, 0, 1)
        for _ in range(3):
            g.multiply_relin(L, n, a, b, ix, o, rescale=True)
        g.sync()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            g.multiply_relin(L, n, a, b, ix, o, rescale=True)
            g.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        res.append(dict(config=f"configs[2] latency mode: CKKS mul+relin+rescale N=2^15 L=16, batch {n}, sync per call", results=n, ms=ms,
                        ops_per_s=n / ms * 1e3))
    g.close()
    return res


def cfg5_bfv_matmul():
    """configs[4]: BFV MatMul 128x128x128 at N=2^15, depth 3 ({60,40,40,60}): 64 row-pair ciphertexts x (BEHZ multiply + relinearize +
    127 rotate_rows(j*128) + add), rotations by non-power-of-two steps through SEAL's NAF terms (355 key switches per ciphertext)."""
    g = be.Context(be.SCHEME_BFV, 32768, bit_sizes=[60, 40, 40, 60], plain_bits=20, device=0)
    L, N, n = g.L, g.N, 64
    a, b = g.alloc(n * 2 * L * N), g.alloc(2 * L * N)
    c3, base, rot, acc = g.alloc(n * 3 * L * N), g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
    g.fill_uniform(a, n * 2 * L, list(range(L)), 1)
    g.fill_uniform(b, 2 * L, list(range(L)), 2)
    g.set_relin_key_synthetic(7)
    k = 0
    while (1 << k) < N // 2:  # the default Galois key set: +-2^k row rotations
        g.set_galois_key_synthetic(g.galois_elt(1 << k), 100 + k)
        g.set_galois_key_synthetic(g.galois_elt(-(1 << k)), 200 + k)
        k += 1
    spacers = (N // 2) // 128
    pw = be.Context.pairwise()

    def run():
        g.bfv_multiply(L, n, a, b, be.Context.outer(0, n, 0, 1), c3)
        g.relinearize(L, n, c3, base)
        cur, nxt = base, acc  # as the bridge's MatMultRow: every rotate + add_inplace pair is one rotate_add pipeline
        for j in range(1, 128):
            g.rotate_add(L, n, base, j * spacers, cur, nxt)
            cur, nxt = (nxt, rot) if cur is base else (nxt, cur)
    ms = timed(g, run, 2)
    out = dict(config="configs[4] BFV MatMul 128x128x128 N=2^15 L=3: 64 row-pair cts x (multiply, relinearize, 127 rotate_rows + add)", results=n, ms=ms,
               latency_ms_per_matrix_product=ms, key_switches_total=64 * 356)
    g.close()
    return out


def client_side():
    """Device-side encryption / decryption rates at the headline parameters (N=2^15, 17 key primes), inputs resident in HBM.
    Keys are uniform residues (the rates do not depend on the key values); the product only — no oracle here."""
    import numpy as np
    g = be.Context(be.SCHEME_CKKS, 32768, bit_sizes=be.chain_bits(16, 45), device=0)
    L, N, K, n = g.L, g.N, g.K, 256
    rng = np.random.default_rng(0)
    pk = np.stack([rng.integers(0, q, (2, N), dtype=np.uint64) for q in g.moduli], axis=1)  # [2][K][N]
    sk = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in g.moduli])               # [K][N]
    g.set_public_key(pk)
    g.set_secret_key(sk)
    plain, ct, ph = g.alloc(n * L * N), g.alloc(n * 2 * L * N), g.alloc(n * L * N)
    g.fill_uniform(plain, n * L, list(range(L)), 3)
    ms_e = timed(g, lambda: g.encrypt(n, plain, 5, 0, ct), 3)
    ms_d = timed(g, lambda: g.decrypt(L, 2, n, ct, ph), 5)
    vals = g.to_device(rng.uniform(-1, 1, (n, N // 2)).view(np.uint64))
    ms_c = timed(g, lambda: g.ckks_encode(n, vals, N // 2, 2.0 ** 40, plain), 3)
    out = g.alloc(n * (N // 2))
    ms_dc = timed(g, lambda: g.ckks_decode(L, n, plain, 2.0 ** 40, out), 3)
    g.keygen_relin(1)
    g.sync()
    ms_k = timed(g, lambda: g.keygen_relin(1), 2)
    g.close()
    return [dict(config="device encrypt (he355_encrypt) CKKS N=2^15 L=16, 256 plaintexts resident", results=n, ms=ms_e, ops_per_s=n / ms_e * 1e3),
            dict(config="device decrypt (he355_decrypt, size 2) CKKS N=2^15 L=16", results=n, ms=ms_d, ops_per_s=n / ms_d * 1e3),
            dict(config="device CKKS encode (he355_ckks_encode, 16384 slots) N=2^15 L=16", results=n, ms=ms_c, ops_per_s=n / ms_c * 1e3),
            dict(config="device CKKS decode (he355_ckks_decode) N=2^15 L=16", results=n, ms=ms_dc, ops_per_s=n / ms_dc * 1e3),
            dict(config="device relinearization-key generation (he355_keygen_relin, 136 MiB key) N=2^15 L=16", results=1, ms=ms_k, ops_per_s=1e3 / ms_k)]


if __name__ == "__main__":
    groups = {"cfg0": lambda: [cfg1_bfv_add()], "cfg1": cfg2_ckks_multiply, "cfg3": lambda: [cfg4_dot_product()], "cfg4": lambda: [cfg5_bfv_matmul()],
              "latency": headline_latency, "client": client_side}
    for name in (sys.argv[1:] or list(groups)):  # no arguments: everything
        for r in groups[name]():
            print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
