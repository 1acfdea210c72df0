"""Client side on the device: he355_encrypt / he355_decrypt against the oracle, bit-exact.
Encryption randomness is counter-based and shared (csrc/client/sampler.h; numpy mirror tests/sampler_np.py), so the
oracle is driven with exactly the polynomials the device sampled (ho_encrypt_explicit)."""
import importlib

import numpy as np
import pytest

import sampler_np as sn

pytestmark = pytest.mark.gpu

CASES = {
    # name: (scheme, N, bit sizes, plain bits)
    "ckks_n2048_mixed": ("ckks", 2048, [60, 40, 40, 60], 0),
    "ckks_n8192_default": ("ckks", 8192, [60, 45, 60], 0),
    "ckks_n32768_d4": ("ckks", 32768, [60, 45, 45, 45, 60], 0),
    "ckks_n4096_single_prime": ("ckks", 4096, [50], 0),
    "bfv_n2048": ("bfv", 2048, [50, 40, 50], 20),
    "bfv_n8192_default": ("bfv", 8192, [60, 40, 60], 20),
    "bfv_n16384_d3": ("bfv", 16384, [60, 40, 40, 60], 20),
    # 17 data primes: more than the 16 the device-side BFV decryptor / BEHZ multiply support -- encryption must still be exact
    # (round 1 overran a 16-entry table here); decryption of such a context runs on the host client
    "bfv_n32768_17_data_primes": ("bfv", 32768, [60] + [45] * 16 + [60], 20),
}


@pytest.fixture(scope="module")
def be():
    mod = importlib.import_module("reference-seal-backend_amd")
    if mod.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run on the MI355X box")
    return mod


@pytest.fixture(params=list(CASES))
def case(request, be, oracle):
    scheme, N, bits, pb = CASES[request.param]
    sid_g = be.SCHEME_CKKS if scheme == "ckks" else be.SCHEME_BFV
    sid_o = oracle.SCHEME_CKKS if scheme == "ckks" else oracle.SCHEME_BFV
    g = be.Context(sid_g, N, bit_sizes=bits, plain_bits=pb, sec128=False, device=0)
    o = oracle.Context(sid_o, N, bit_sizes=bits, plain_bits=pb, sec128=False)
    assert g.moduli == o.moduli
    sk = o.keygen_secret(11)
    pk = o.keygen_public(sk, 12)
    g.set_public_key(pk)
    g.set_secret_key(sk)
    yield scheme, g, o, sk, pk, np.random.default_rng(len(request.param))
    g.close()


def test_encrypt_matches_oracle(case, be):
    scheme, g, o, sk, pk, rng = case
    N, L = g.N, g.L
    n, seed, first = 35, 0xC0FFEE1234, 1000  # more than one internal chunk (32), ragged
    if scheme == "ckks":
        plains = np.stack([o.random_poly(rng, L, 1)[0] for _ in range(n)])  # [n, L, N] NTT-form plaintexts
    else:
        plains = rng.integers(0, o.t, (n, N)).astype(np.uint64)
    dp = g.to_device(plains)
    out = g.alloc(n * 2 * L * N)
    g.encrypt(n, dp, seed, first, out)
    got = out.download((n, 2, L, N))
    for r in (0, 1, 31, 32, 34):
        su, s0, s1 = sn.enc_streams(first + r)
        want = o.encrypt_explicit(pk, plains[r], sn.sample_ternary(seed, su, N), sn.sample_cbd(seed, s0, N), sn.sample_cbd(seed, s1, N))
        assert np.array_equal(got[r], want), r
    # and they decrypt: the oracle's phase of a device ciphertext carries the plaintext (plus small noise)
    ph = o.decrypt_phase(got[3], sk)
    if scheme == "bfv":
        assert np.array_equal(o.bfv_decode_phase(ph), plains[3])
    else:
        q0 = o.moduli[0]
        diff = ((ph[0].astype(object) - plains[3][0].astype(object)) % q0).astype(np.uint64)  # e0 + e1*s + u*e under prime 0, NTT form
        c0 = o.intt(0, diff).astype(object)
        c0 = np.where(c0 > q0 // 2, c0 - q0, c0)
        assert max(abs(int(v)) for v in c0) < 2 ** 14  # the encryption noise stays tiny


def test_decrypt_matches_oracle(case, be):
    scheme, g, o, sk, pk, rng = case
    N = g.N
    if scheme == "bfv" and g.L > 16:
        with pytest.raises(be.HE355Error):  # loud, not wrong: the bridge decrypts such contexts with the host client
            g.decrypt(g.L, 2, 1, g.alloc(2 * g.L * N), g.alloc(N))
        return
    for L in sorted({g.L, max(1, g.L - 1)}):
        for size in (2, 3):
            n = 3
            cts = np.stack([o.random_poly(rng, L, size) for _ in range(n)])  # any residues: the phase is a function of ct and sk
            d = g.to_device(cts)
            if scheme == "ckks":
                out = g.alloc(n * L * N)
                g.decrypt(L, size, n, d, out)
                got = out.download((n, L, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.decrypt_phase(cts[r], sk)), (L, size, r)
            else:
                out = g.alloc(n * N)
                g.decrypt(L, size, n, d, out)
                got = out.download((n, N))
                for r in range(n):
                    assert np.array_equal(got[r], o.bfv_decode_phase(o.decrypt_phase(cts[r], sk))), (L, size, r)


def test_round_trip_through_the_evaluator(case, be):
    """Enc on the device -> add on the device -> Dec on the device gives the sum of the plaintexts (BFV exactly; CKKS up
    to the encryption noise, checked in coefficient form)."""
    scheme, g, o, sk, pk, rng = case
    N, L = g.N, g.L
    if scheme == "bfv" and L > 16:
        pytest.skip("device-side BFV decryption supports up to 16 data primes")
    if scheme == "bfv":
        a, b = rng.integers(0, o.t, (2, N)).astype(np.uint64)
        d = g.to_device(np.stack([a, b]))
        cts, s, out = g.alloc(2 * 2 * L * N), g.alloc(2 * L * N), g.alloc(N)
        g.encrypt(2, d, 77, 0, cts)
        g.add(L, 2, 1, cts, cts, be.Context.outer(0, 1, 1, 1), s)
        g.decrypt(L, 2, 1, s, out)
        assert np.array_equal(out.download((N,)), (a + b) % np.uint64(o.t))
    else:
        scale = 2.0 ** 30
        x = rng.uniform(-1, 1, N // 2)
        import oracle as ho
        plain = ho.ckks_encode(o, x, scale)
        d = g.to_device(np.ascontiguousarray(plain[None]))
        ct, out = g.alloc(2 * L * N), g.alloc(L * N)
        g.encrypt(1, d, 78, 5, ct)
        g.decrypt(L, 2, 1, ct, out)
        assert np.allclose(ho.ckks_decode(o, out.download((L, N)), scale).real, x, atol=1e-4)


def test_encrypt_zero_and_accumulate_count_zero(case, be):
    """accumulateCKKS / accumulateBFV with count == 0 return encryptor()->encrypt_zero (/root/reference/src/engine/
    seal_context.cpp:312-316, 341-344): a FRESH encryption of zero at the first data level replaces each ciphertext.
    he355_accumulate(count = 0) draws it from the context's zero stream; with the stream pinned the result is the oracle's
    encryption of the zero plaintext under the same sampled polynomials, bit for bit, and it decrypts to zero."""
    scheme, g, o, sk, pk, rng = case
    N, L = g.N, g.L
    n, seed, first = 3, 0xABCDEF, 40
    zero_plain = np.zeros((L, N), dtype=np.uint64) if scheme == "ckks" else np.zeros(N, dtype=np.uint64)
    want = []
    for r in range(n):
        su, s0, s1 = sn.enc_streams(first + r)
        want.append(o.encrypt_explicit(pk, zero_plain, sn.sample_ternary(seed, su, N), sn.sample_cbd(seed, s0, N), sn.sample_cbd(seed, s1, N)))
    ez = g.alloc(n * 2 * L * N)
    g.encrypt_zero(n, seed, first, ez)
    got = ez.download((n, 2, L, N))
    for r in range(n):
        assert np.array_equal(got[r], want[r]), r
    # accumulate(count = 0): whatever the slab held is replaced
    junk = np.stack([o.random_poly(rng, L, 2) for _ in range(n)])
    slab, tmp = g.to_device(junk), g.alloc(n * 2 * L * N)
    g.set_zero_stream(seed, first)
    g.accumulate(L, n, slab, 0, tmp)
    acc = slab.download((n, 2, L, N))
    for r in range(n):
        assert np.array_equal(acc[r], want[r]), r
    # a second call continues the stream: fresh randomness, still encryptions of zero
    g.accumulate(L, n, slab, 0, tmp)
    acc2 = slab.download((n, 2, L, N))
    assert not np.array_equal(acc2, acc)
    ph = o.decrypt_phase(acc2[1], sk)
    if scheme == "bfv":
        assert not o.bfv_decode_phase(ph).any()
    else:
        q0 = o.moduli[0]
        c0 = o.intt(0, ph[0]).astype(object)
        c0 = np.where(c0 > q0 // 2, c0 - q0, c0)
        assert max(abs(int(v)) for v in c0) < 2 ** 14
    if L > 1:  # SEAL returns a top-level ciphertext there: a lower-level slab cannot hold it
        with pytest.raises(be.HE355Error):
            g.accumulate(L - 1, n, slab, 0, tmp)


# ---- encoders on the device ---------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sim():
    import ctypes as C
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run(["make", "-C", os.path.join(here, "csim"), "-s"], check=True)
    L = C.CDLL(os.path.join(here, "csim", "_build", "libcsim.so"))
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.sim_params_create.restype = vp
    L.sim_params_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_int), C.c_size_t, C.c_int, C.c_int]
    L.sim_params_destroy.argtypes = [vp]
    L.simc_create.restype = vp
    L.simc_create.argtypes = [vp, C.c_uint64]
    L.simc_destroy.argtypes = [vp]
    L.simc_ckks_encode.argtypes = [vp, C.POINTER(C.c_double), C.c_size_t, C.c_double, u64p]
    L.simc_ckks_decode.argtypes = [vp, u64p, C.c_size_t, C.c_double, C.POINTER(C.c_double)]
    L.simc_bfv_encode.argtypes = [vp, C.POINTER(C.c_int64), C.c_size_t, u64p]
    L.simc_bfv_decode.argtypes = [vp, u64p, C.POINTER(C.c_int64)]
    L.simc_secret_key.argtypes = [vp, u64p]
    L.simc_relin_key.argtypes = [vp, u64p]
    L.simc_galois_key.argtypes = [vp, C.c_uint32, u64p]
    L.simc_keygen_seed.restype = C.c_uint64
    L.simc_keygen_seed.argtypes = [vp]
    return L


def _host_client(sim, scheme, N, bits, pb):
    import ctypes as C
    arr = (C.c_int * len(bits))(*bits)
    p = sim.sim_params_create(2 if scheme == "ckks" else 1, N, arr, len(bits), pb, 0)
    return p, sim.simc_create(p, 42)


def test_encoders_match_host_client_and_oracle(request, case, be, sim, oracle):
    """he355_ckks_encode / _decode and he355_bfv_encode / _decode: bit-identical to the product's host encoders (same inline
    floating-point code and tables: csrc/client/ckks_codec.h) and consistent with the oracle's numpy restatement (CKKS: the
    encoders are floating point; the coefficient rounding alone leaves ~sqrt(N)/2/scale per slot, 8e-8 at N=2^15 and scale 2^30:
    tolerance 1e-6; BFV: exact)."""
    import ctypes as C
    import oracle as ho
    scheme, g, o, sk, pk, rng = case
    N, L = g.N, g.L
    name = request.node.callspec.params["case"]
    _, _, bits, pb = CASES[name]
    p, c = _host_client(sim, scheme, N, bits, pb)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    if scheme == "ckks":
        scale, n = 2.0 ** 30, 3
        for count in (N // 2, 5):
            x = rng.uniform(-1, 1, (n, count))
            dv = g.to_device(x.view(np.uint64))
            plain = g.alloc(n * L * N)
            g.ckks_encode(n, dv, count, scale, plain)
            got = plain.download((n, L, N))
            for r in range(n):
                want = np.empty((L, N), dtype=np.uint64)
                row = np.ascontiguousarray(x[r])
                sim.simc_ckks_encode(c, row.ctypes.data_as(dp), count, scale, oracle._p(want))
                assert np.array_equal(got[r], want), (count, r)
                assert np.allclose(ho.ckks_decode(o, got[r], scale).real[:count], x[r], atol=1e-6)
            # decode on the device: bit-identical to the host decoder, and the values come back
            for Ld in sorted({L, max(1, L - 1)}):
                sub = np.ascontiguousarray(got[:, :Ld])
                dpl = g.to_device(sub)
                out = g.alloc(n * (N // 2))
                g.ckks_decode(Ld, n, dpl, scale, out)
                vals = out.download().view(np.float64).reshape(n, N // 2)
                for r in range(n):
                    want = np.empty(N // 2)
                    sim.simc_ckks_decode(c, oracle._p(np.ascontiguousarray(sub[r])), Ld, scale, want.ctypes.data_as(dp))
                    assert np.array_equal(vals[r], want), (count, Ld, r)
                    assert np.allclose(vals[r, :count], x[r], atol=1e-6) and np.allclose(vals[r, count:], 0, atol=1e-6)
                # he355_ckks_decode_slots: only the slots a workload's decode() reads -- the same bits as those slots of the full decode
                for ranges in ([(0, 1)], [(0, 5)], [(3, 4), (N // 2 - 2, 2)], [(0, N // 2)], [(0, 2), (7, 1), (N // 4, 3), (1, 2)]):
                    tot = sum(cnt for _, cnt in ranges)
                    outs = g.alloc(n * tot)
                    g.ckks_decode_slots(Ld, n, dpl, scale, ranges, outs)
                    part = outs.download().view(np.float64).reshape(n, tot)
                    assert np.array_equal(part, np.concatenate([vals[:, f:f + cnt] for f, cnt in ranges], axis=1)), (Ld, ranges)
                with pytest.raises(be.HE355Error):
                    g.ckks_decode_slots(Ld, n, dpl, scale, [(N // 2 - 1, 2)], out)  # beyond the encoder's slots
    else:
        n = 3
        codec = ho.BatchCodec(N, o.t)
        for count in (N, 7):
            x = rng.integers(-(o.t // 2), o.t // 2, (n, count)).astype(np.int64)
            dv = g.to_device(x.view(np.uint64))
            plain = g.alloc(n * N)
            g.bfv_encode(n, dv, count, plain)
            got = plain.download((n, N))
            for r in range(n):
                full = np.zeros(N, dtype=np.int64)
                full[:count] = x[r]
                assert np.array_equal(got[r], codec.encode(full)), (count, r)
                want = np.empty(N, dtype=np.uint64)
                sim.simc_bfv_encode(c, np.ascontiguousarray(x[r]).ctypes.data_as(ip), count, oracle._p(want))
                assert np.array_equal(got[r], want)
            out = g.alloc(n * N)
            g.bfv_decode(n, plain, out)
            vals = out.download().view(np.int64).reshape(n, N)
            assert np.array_equal(vals[:, :count], x) and not vals[:, count:].any()
            # he355_bfv_decode_slots: e.g. the first dim3 slots of both batching rows (the row-major product's decode, bfv row .cpp:339-369)
            for ranges in ([(0, 1)], [(0, 7), (N // 2, 7)], [(N - 3, 3)], [(0, N)], [(5, 2), (0, 1), (N // 2 + 1, 4), (9, 9)]):
                tot = sum(cnt for _, cnt in ranges)
                outs = g.alloc(n * tot)
                g.bfv_decode_slots(n, plain, ranges, outs)
                part = outs.download().view(np.int64).reshape(n, tot)
                assert np.array_equal(part, np.concatenate([vals[:, f:f + cnt] for f, cnt in ranges], axis=1)), ranges
            with pytest.raises(be.HE355Error):
                g.bfv_decode_slots(n, plain, [(N, 1)], out)
    sim.simc_destroy(c)
    sim.sim_params_destroy(p)


def test_device_keygen_equals_host_keygen(request, case, be, sim, oracle):
    """he355_keygen_relin / he355_keygen_galois from the host client's secret key and key seed: relinearization and rotation
    with the device-generated keys are bit-identical to the same operations with the host-generated keys uploaded (random
    inputs: equal outputs for every digit and residue <=> equal keys), and the keys work (oracle decryption of a rotation)."""
    scheme, g, o, sk, pk, rng = case
    if g.K < 2:
        pytest.skip("no key switching with a single prime")
    N, L, K = g.N, g.L, g.K
    name = request.node.callspec.params["case"]
    _, _, bits, pb = CASES[name]
    p, c = _host_client(sim, scheme, N, bits, pb)
    hsk = np.empty((K, N), dtype=np.uint64)
    sim.simc_secret_key(c, oracle._p(hsk))
    seed = sim.simc_keygen_seed(c)
    g.set_secret_key(hsk)
    elt = g.galois_elt(1) if scheme == "ckks" else 3
    n = 2
    ct3, ct2 = np.stack([o.random_poly(rng, L, 3) for _ in range(n)]), np.stack([o.random_poly(rng, L, 2) for _ in range(n)])
    d3, d2 = g.to_device(ct3), g.to_device(ct2)
    outs = {}
    for who in ("device", "host"):
        if who == "device":
            g.keygen_relin(seed)
            g.keygen_galois(elt, seed)
        else:
            rk, gk = np.empty((L, 2, K, N), dtype=np.uint64), np.empty((L, 2, K, N), dtype=np.uint64)
            sim.simc_relin_key(c, oracle._p(rk))
            sim.simc_galois_key(c, elt, oracle._p(gk))
            g.set_relin_key(rk)
            g.set_galois_key(elt, gk)
        a, b = g.alloc(n * 2 * L * N), g.alloc(n * 2 * L * N)
        g.relinearize(L, n, d3, a)
        g.apply_galois(L, n, d2, elt, b)
        outs[who] = (a.download((n, 2, L, N)), b.download((n, 2, L, N)))
    assert np.array_equal(outs["device"][0], outs["host"][0]) and np.array_equal(outs["device"][1], outs["host"][1])
    # with the host key in the oracle the same results come out, so the device key is a valid key for this secret key
    assert np.array_equal(outs["device"][0][0], o.relinearize(ct3[0], rk))
    sim.simc_destroy(c)
    sim.sim_params_destroy(p)
