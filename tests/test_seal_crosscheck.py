"""OPT-IN cross-check against a real Microsoft SEAL install (SEAL_INSTALL_DIR, the variable the reference's build uses:
/root/reference/cmake/utils/import-library.cmake:54-58).  SEAL is not in this image, so these tests are SKIPPED here and on the
GPU box; where SEAL v3.7.x is installed they build tools/seal_crosscheck, let real SEAL generate keys, encrypt and evaluate, and
hold the oracle (CPU) and the HIP path (GPU) to SEAL's outputs bit for bit.  Until that has run somewhere, parity against SEAL
itself stays unpinned; the exact big-integer model (tests/test_exact_model.py) is the second source in the
meantime."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEAL_DIR = os.environ.get("SEAL_INSTALL_DIR", "")
pytestmark = pytest.mark.skipif(not SEAL_DIR, reason="SEAL_INSTALL_DIR not set: Microsoft SEAL is not available offline")

CASES = [("ckks", 8192, 2, 45, 45), ("ckks", 16384, 4, 45, 45), ("ckks", 32768, 16, 45, 45),  # the last: the headline chain {60, 45 x 15, 60}
         ("bfv", 8192, 2, 40, 20), ("bfv", 16384, 3, 40, 20),
         ("bfv", 32768, 3, 40, 20)]  # the last: BASELINE configs[4]'s chain {60, 40, 40, 60}, t = 786433 (exact_vectors_big.json: bfv_n32768_60_40_40_60)


def _pinned(meta, who):
    """one line per case a run has pinned against SEAL itself (shown with pytest -s): what the first contact settled"""
    ckks = meta["scheme"] == "ckks"
    ops = "primes, add, multiply, relinearize, rescale, rotate(1), rotate(3 = -1 + 4)" if ckks else \
          "primes, t, add, BEHZ multiply, relinearize, rotate_rows(1), rotate_rows(3 = -1 + 4)"
    print(f"PINNED against SEAL 3.7 ({who}): {meta['scheme']} N={meta['N']} depth={meta['depth']} bits={meta['coeff_bits']}: {ops}")
    if ckks:
        big = " (the headline chain; exact_vectors_big.json: ckks_n32768_60_45x15_60)" if meta["N"] == 32768 and meta["depth"] == 16 else ""
        print("  -> also pins what the exact model only derives: minimal-root choice, digit/key layout, floor directions" + big)
    else:
        print("  -> pins SURVEY.md App. A's recollection of BEHZ: m~ = 2^32, uncorrected fast base conversions, auxiliary base choice")


def _run_seal(tmp_path, case):
    tool_dir = os.path.join(ROOT, "tools", "seal_crosscheck")
    subprocess.run(["make", "-C", tool_dir, "-s", f"SEAL_INSTALL_DIR={SEAL_DIR}"], check=True)
    out = tmp_path / "_".join(str(c) for c in case)
    out.mkdir()
    subprocess.run([os.path.join(tool_dir, "_build", "seal_crosscheck"), *[str(c) for c in case], str(out)], check=True)
    meta = json.load(open(out / "meta.json"))
    N, K = meta["N"], len(meta["primes"])
    L = K - 1

    def ct(name, size, lvl=L):
        return np.fromfile(out / f"{name}.bin", dtype=np.uint64).reshape(size, lvl, N)

    def key(name):
        return np.fromfile(out / f"{name}.bin", dtype=np.uint64).reshape(L, 2, K, N)
    return meta, ct, key, L


def _check(ops, meta, ct, key, L):
    ckks = meta["scheme"] == "ckks"
    a, b = ct("a", 2), ct("b", 2)
    rk = key("relin")
    ge = {int(s): int(e) for s, e in meta["galois_elts"].items()}
    gk = {e: key(f"galois_{e}") for e in ge.values()}
    assert np.array_equal(ops.add(a, b), ct("out_add", 2))
    c3 = ops.multiply(a, b)
    assert np.array_equal(c3, ct("out_multiply", 3))
    rl = ops.relinearize(c3, rk)
    assert np.array_equal(rl, ct("out_multiply_relin", 2))
    if ckks:
        assert np.array_equal(ops.rescale(rl), ct("out_multiply_relin_rescale", 2, L - 1))
    assert np.array_equal(ops.apply_galois(a, ge[1], gk[ge[1]]), ct("out_rotate_1", 2))
    assert np.array_equal(ops.apply_galois(ops.apply_galois(a, ge[-1], gk[ge[-1]]), ge[4], gk[ge[4]]), ct("out_rotate_3", 2))


@pytest.mark.parametrize("case", CASES)
def test_oracle_against_real_seal(oracle, tmp_path, case):
    meta, ct, key, L = _run_seal(tmp_path, case)
    sid = oracle.SCHEME_CKKS if meta["scheme"] == "ckks" else oracle.SCHEME_BFV
    o = oracle.Context(sid, meta["N"], primes=meta["primes"], plain_modulus=meta["plain_modulus"])
    # the reference's parameter rule, restated, must pick SEAL's primes
    bits = [60] + [meta["coeff_bits"]] * (meta["depth"] - 1) + [60]
    o2 = oracle.Context(sid, meta["N"], bit_sizes=bits, plain_bits=meta["extra_bits"] if meta["scheme"] == "bfv" else 0)
    assert [int(q) for q in o2.moduli] == meta["primes"] and int(o2.t) == meta["plain_modulus"]

    class Ops:
        add = staticmethod(o.add)
        relinearize = staticmethod(o.relinearize)
        rescale = staticmethod(o.rescale)
        apply_galois = staticmethod(o.apply_galois)

        @staticmethod
        def multiply(a, b):
            return o.multiply_ntt(a, b) if meta["scheme"] == "ckks" else o.bfv_multiply(a, b)
    _check(Ops, meta, ct, key, L)
    _pinned(meta, "oracle, CPU")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_path_against_real_seal(tmp_path, case):
    be = importlib.import_module("reference-seal-backend_amd")
    meta, ct, key, L = _run_seal(tmp_path, case)
    ckks = meta["scheme"] == "ckks"
    N = meta["N"]
    g = be.Context(be.SCHEME_CKKS if ckks else be.SCHEME_BFV, N, primes=meta["primes"], plain_modulus=meta["plain_modulus"], device=0)
    pw = be.Context.pairwise()
    g.set_relin_key(key("relin"))
    for e in meta["galois_elts"].values():
        g.set_galois_key(int(e), key(f"galois_{e}"))

    class Ops:
        @staticmethod
        def add(a, b):
            out = g.alloc(a.size)
            g.add(a.shape[1], 2, 1, g.to_device(a[None]), g.to_device(b[None]), pw, out)
            return out.download(a.shape)

        @staticmethod
        def multiply(a, b):
            out = g.alloc(3 * a.shape[1] * N)
            (g.multiply if ckks else g.bfv_multiply)(a.shape[1], 1, g.to_device(a[None]), g.to_device(b[None]), pw, out)
            return out.download((3, a.shape[1], N))

        @staticmethod
        def relinearize(c3, rk):
            out = g.alloc(2 * c3.shape[1] * N)
            g.relinearize(c3.shape[1], 1, g.to_device(c3[None]), out)
            return out.download((2, c3.shape[1], N))

        @staticmethod
        def rescale(c):
            out = g.alloc(2 * (c.shape[1] - 1) * N)
            g.rescale(c.shape[1], 2, 1, g.to_device(c[None]), out)
            return out.download((2, c.shape[1] - 1, N))

        @staticmethod
        def apply_galois(c, elt, k):
            out = g.alloc(c.size)
            g.apply_galois(c.shape[1], 1, g.to_device(c[None]), elt, out)
            return out.download(c.shape)
    _check(Ops, meta, ct, key, L)
    _pinned(meta, "HIP path, MI355X")
    g.close()
