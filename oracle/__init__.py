"""CPU oracle bindings — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``reference-seal-backend_amd/``) never does.  See ``oracle/he_oracle.h`` for what
the oracle restates (SEAL v3.7.2 behind the reference's ``seal::Evaluator`` calls) and for the
"parity unpinned" statement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HE_ORACLE_LIB_PATH: another build of the same oracle (the ASan/UBSan build of `make -C oracle asan`, tools/asan_oracle.sh)
_LIB_PATH = os.environ.get("HE_ORACLE_LIB_PATH") or os.path.join(_HERE, "_build", "libhe_oracle.so")

SCHEME_BFV = 1
SCHEME_CKKS = 2
OP_ADD, OP_MUL, OP_MUL_RELIN, OP_MUL_RELIN_RESCALE, OP_DOT = 0, 1, 2, 3, 4


def _host_signature() -> str:
    """what `-march=native` depends on: the CPU model and its feature flags"""
    model, flags = "", ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and not model:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = " ".join(sorted(line.split(":", 1)[1].split()))
                if model and flags:
                    break
    except OSError:
        pass
    import hashlib
    return model + " " + hashlib.sha256(flags.encode()).hexdigest()[:16]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it.
    The library is built `-march=native`, so it belongs to the host that built it: `_build/host.sig` records that host's CPU
    model + flags and a different host (the GPU box receives the build container's `_build/`) rebuilds before the first use -- the
    timed CPU baseline then runs code tuned for the cores it is timed on, and never code the host cannot execute.  A file lock
    serialises concurrent builders (N bench ranks, parallel test workers)."""
    if os.environ.get("HE_ORACLE_LIB_PATH"):
        return _LIB_PATH
    import fcntl
    srcs = [os.path.join(_HERE, f) for f in ("he_oracle.c", "he_oracle_bfv.inc", "he_oracle_pipelines.inc", "he_oracle.h", "Makefile")]
    sig_path = os.path.join(_HERE, "_build", "host.sig")
    sig = _host_signature()

    def stale():
        if force or not os.path.exists(_LIB_PATH):
            return True
        if any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
            return True
        try:
            return open(sig_path).read().strip() != sig
        except OSError:
            return True

    if stale():
        try:
            os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
            with open(os.path.join(_HERE, "_build", ".lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    if stale():  # somebody else may have built it while we waited
                        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
                        with open(sig_path, "w") as f:
                            f.write(sig + "\n")
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        except (subprocess.CalledProcessError, OSError) as e:
            # The tree is read-only, or the rebuild failed.  A library that is present but was built `-march=native` on ANOTHER host may
            # hold instructions this host cannot execute: it is never loaded.  Fall back to a portable build (x86-64-v2) outside the tree.
            _portable_fallback(srcs, e)
    return _LIB_PATH


portable_build = False  # True: the library in use is the x86-64-v2 fallback (bench.py records it next to the CPU baseline)


def _portable_fallback(srcs, why):
    global _LIB_PATH, portable_build
    import sys
    import tempfile
    out_dir = os.path.join(tempfile.gettempdir(), f"he_oracle_{os.getuid()}")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libhe_oracle_portable.so")
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        tmp = out + f".{os.getpid()}.tmp"
        subprocess.run(["gcc", "-O3", "-march=x86-64-v2", "-fopenmp", "-fPIC", "-std=c11", "-shared", "-o", tmp, os.path.join(_HERE, "he_oracle.c"), "-lm"], check=True)
        os.replace(tmp, out)
    print(f"oracle: build in the tree failed ({why}); using a portable build at {out}", file=sys.stderr)
    _LIB_PATH = out
    portable_build = True
    return _LIB_PATH


def effective_cpus() -> dict:
    """The CPU share this process really has: the affinity mask, and the cgroup quota when one is set (a container limited to 16
    CPUs still shows every core of the host in os.cpu_count() and in omp_get_max_threads())."""
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_cpu_max": None,
            "quota_cpus": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().strip()
        except OSError:
            continue
        info["cgroup_cpu_max"] = txt
        try:
            if path.endswith("cpu.max"):
                q, per = txt.split()
                if q != "max":
                    info["quota_cpus"] = float(q) / float(per)
            else:
                q = float(txt)
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    info["quota_cpus"] = q / per
        except (ValueError, OSError):
            pass
        break
    eff = info["affinity"] or info["nproc"] or 1
    if info["quota_cpus"]:
        eff = max(1, min(eff, int(info["quota_cpus"] + 0.5)))
    info["effective"] = eff
    return info


_lib = None
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_u64p)


def _p32(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u32p)


def lib():
    global _lib
    if _lib is None:
        build()  # no-op when the library is present, newer than its sources and was built on this host
        L = C.CDLL(_LIB_PATH)
        vp, sz, u64, i32, u32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32
        sig = {
            "ho_ctx_create": (vp, [i32, sz, C.POINTER(C.c_int), sz, i32, i32, C.c_char_p, sz]),
            "ho_ctx_create_primes": (vp, [i32, sz, _u64p, sz, u64, C.c_char_p, sz]),
            "ho_ctx_destroy": (None, [vp]),
            "ho_scratch_release": (None, []), "ho_scratch_release_all": (None, []),
            "ho_N": (sz, [vp]), "ho_key_mod_count": (sz, [vp]), "ho_data_mod_count": (sz, [vp]),
            "ho_modulus": (u64, [vp, sz]), "ho_plain_modulus": (u64, [vp]), "ho_root": (u64, [vp, sz]),
            "ho_root_powers": (None, [vp, sz, _u64p]),
            "ho_is_prime": (i32, [u64]), "ho_get_primes": (sz, [u64, i32, sz, _u64p]),
            "ho_ntt": (None, [vp, sz, _u64p]), "ho_intt": (None, [vp, sz, _u64p]),
            "ho_add": (None, [vp, sz, sz, _u64p, _u64p, _u64p]),
            "ho_sub": (None, [vp, sz, sz, _u64p, _u64p, _u64p]),
            "ho_multiply_ntt": (None, [vp, sz, _u64p, _u64p, _u64p]),
            "ho_switch_key": (None, [vp, sz, _u64p, _u64p, _u64p]),
            "ho_relinearize": (None, [vp, sz, _u64p, _u64p]),
            "ho_encrypt_explicit": (None, [vp, _u64p, _u64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _u64p]),
            "ho_multiply_plain": (None, [vp, sz, sz, _u64p, _u64p, _u64p]),
            "ho_add_plain": (None, [vp, sz, sz, _u64p, _u64p, _u64p]),
            "ho_mod_switch_drop": (None, [vp, sz, sz, sz, _u64p, _u64p]),
            "ho_rescale": (None, [vp, sz, sz, _u64p, _u64p]),
            "ho_mod_switch_coeff": (None, [vp, sz, sz, _u64p, _u64p]),
            "ho_galois_elt_from_step": (u32, [vp, i32]),
            "ho_galois_elts_all": (sz, [vp, _u32p]),
            "ho_apply_galois_poly": (None, [vp, sz, u32, i32, _u64p, _u64p]),
            "ho_apply_galois": (None, [vp, sz, u32, _u64p, _u64p, _u64p]),
            "ho_bfv_multiply": (None, [vp, sz, _u64p, _u64p, _u64p]),
            "ho_batch_op": (None, [vp, i32, sz, sz, _u64p, _u32p, _u64p, _u32p, _u64p, _u64p, i32]),
            "ho_max_threads": (i32, []),
            "ho_rotate": (i32, [vp, sz, i32, sz, _u32p, C.POINTER(_u64p), _u64p, _u64p]),
            "ho_accumulate": (i32, [vp, sz, sz, sz, _u32p, C.POINTER(_u64p), _u64p]),
            "ho_batch_outer": (i32, [vp, i32, sz, sz, sz, _u64p, _u64p, _u64p, sz, _u32p, C.POINTER(_u64p), sz, _u64p, i32]),
            "ho_bfv_matmul_rows": (i32, [vp, sz, sz, _u64p, _u64p, _u64p, sz, _u32p, C.POINTER(_u64p), sz, _u64p, i32]),
            "ho_keygen_secret": (None, [vp, u64, _u64p]),
            "ho_keygen_public": (None, [vp, _u64p, u64, _u64p]),
            "ho_keygen_kswitch": (None, [vp, _u64p, _u64p, u64, _u64p]),
            "ho_keygen_relin": (None, [vp, _u64p, u64, _u64p]),
            "ho_keygen_galois": (None, [vp, _u64p, u32, u64, _u64p]),
            "ho_encrypt": (None, [vp, _u64p, _u64p, u64, _u64p]),
            "ho_decrypt_phase": (None, [vp, sz, sz, _u64p, _u64p, _u64p]),
            "ho_bfv_decode_phase": (None, [vp, sz, _u64p, _u64p]),
            "ho_crt_to_double": (None, [vp, sz, _u64p, C.c_double, C.POINTER(C.c_double)]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def chain_bits(depth: int, coeff_bits: int) -> list[int]:
    """The reference's parameter rule {60, bits x (depth-1), 60} (seal_context.cpp:79-82,107-110)."""
    return [60] + [coeff_bits] * (depth - 1) + [60]


class Context:
    """Thin object wrapper over ho_ctx."""

    def __init__(self, scheme: int, N: int, bit_sizes=None, primes=None, plain_bits: int = 0,
                 plain_modulus: int = 0, sec128: bool = True):
        L = lib()
        err = C.create_string_buffer(256)
        if primes is not None:
            arr = np.asarray(primes, dtype=np.uint64)
            self.h = L.ho_ctx_create_primes(scheme, N, _p(arr), len(arr), plain_modulus, err, 256)
        else:
            bs = (C.c_int * len(bit_sizes))(*bit_sizes)
            self.h = L.ho_ctx_create(scheme, N, bs, len(bit_sizes), plain_bits, int(sec128), err, 256)
        if not self.h:
            raise ValueError(err.value.decode())
        self.scheme = scheme
        self.N = N
        self.K = L.ho_key_mod_count(self.h)
        self.L = L.ho_data_mod_count(self.h)
        self.moduli = [L.ho_modulus(self.h, i) for i in range(self.K)]
        self.t = L.ho_plain_modulus(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().ho_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------------
    def root(self, i):
        return lib().ho_root(self.h, i)

    def root_powers(self, i):
        out = np.empty(self.N, dtype=np.uint64)
        lib().ho_root_powers(self.h, i, _p(out))
        return out

    def random_poly(self, rng: np.random.Generator, L: int, size: int = 1, special: bool = False) -> np.ndarray:
        """Uniform residues, shape [size, L(+1 if special), N] (throughput-mode synthetic data)."""
        mods = self.moduli[:L] + ([self.moduli[-1]] if special else [])
        out = np.empty((size, len(mods), self.N), dtype=np.uint64)
        for i, q in enumerate(mods):
            out[:, i, :] = rng.integers(0, q, size=(size, self.N), dtype=np.uint64)
        return out

    def random_kswitch_key(self, rng, ) -> np.ndarray:
        """Uniformly random 'key' [Ltop][2][K][N] — enough for bit-exact evaluator parity."""
        out = np.empty((self.L, 2, self.K, self.N), dtype=np.uint64)
        for i, q in enumerate(self.moduli):
            out[:, :, i, :] = rng.integers(0, q, size=(self.L, 2, self.N), dtype=np.uint64)
        return out

    # -- transforms ------------------------------------------------------------------------------
    def ntt(self, i, poly):
        p = np.ascontiguousarray(poly, dtype=np.uint64).copy()
        lib().ho_ntt(self.h, i, _p(p))
        return p

    def intt(self, i, poly):
        p = np.ascontiguousarray(poly, dtype=np.uint64).copy()
        lib().ho_intt(self.h, i, _p(p))
        return p

    # -- evaluator -------------------------------------------------------------------------------
    def add(self, a, b):
        size, L, _ = a.shape
        out = np.empty_like(a)
        lib().ho_add(self.h, L, size, _p(a), _p(b), _p(out))
        return out

    def multiply_ntt(self, a, b):
        L = a.shape[1]
        out = np.empty((3, L, self.N), dtype=np.uint64)
        lib().ho_multiply_ntt(self.h, L, _p(a), _p(b), _p(out))
        return out

    def switch_key(self, target, key, ct):
        L = target.shape[0]
        out = np.ascontiguousarray(ct).copy()
        lib().ho_switch_key(self.h, L, _p(np.ascontiguousarray(target)), _p(key), _p(out))
        return out

    def multiply_plain(self, ct, plain):
        size, L, _ = ct.shape
        out = np.empty_like(ct)
        lib().ho_multiply_plain(self.h, L, size, _p(np.ascontiguousarray(ct)), _p(np.ascontiguousarray(plain)), _p(out))
        return out

    def add_plain(self, ct, plain):
        size, L, _ = ct.shape
        out = np.empty_like(ct)
        lib().ho_add_plain(self.h, L, size, _p(np.ascontiguousarray(ct)), _p(np.ascontiguousarray(plain)), _p(out))
        return out

    def mod_switch_drop(self, x, L_to):
        size, L, _ = x.shape
        out = np.empty((size, L_to, self.N), dtype=np.uint64)
        lib().ho_mod_switch_drop(self.h, L, L_to, size, _p(np.ascontiguousarray(x)), _p(out))
        return out

    def relinearize(self, ct3, rk):
        L = ct3.shape[1]
        t = np.ascontiguousarray(ct3).copy()
        lib().ho_relinearize(self.h, L, _p(t), _p(rk))
        return np.ascontiguousarray(t[:2])

    def rescale(self, ct):
        size, L, _ = ct.shape
        out = np.empty((size, L - 1, self.N), dtype=np.uint64)
        lib().ho_rescale(self.h, L, size, _p(np.ascontiguousarray(ct)), _p(out))
        return out

    def mod_switch_coeff(self, ct):
        size, L, _ = ct.shape
        out = np.empty((size, L - 1, self.N), dtype=np.uint64)
        lib().ho_mod_switch_coeff(self.h, L, size, _p(np.ascontiguousarray(ct)), _p(out))
        return out

    def galois_elt(self, step):
        return lib().ho_galois_elt_from_step(self.h, step)

    def galois_elts_all(self):
        buf = np.zeros(64, dtype=np.uint32)
        n = lib().ho_galois_elts_all(self.h, _p32(buf))
        return [int(x) for x in buf[:n]]

    def apply_galois_poly(self, i, elt, ntt_form, poly):
        out = np.empty(self.N, dtype=np.uint64)
        lib().ho_apply_galois_poly(self.h, i, elt, int(ntt_form), _p(np.ascontiguousarray(poly)), _p(out))
        return out

    def apply_galois(self, ct, elt, gkey):
        L = ct.shape[1]
        out = np.empty_like(ct)
        lib().ho_apply_galois(self.h, L, elt, _p(gkey), _p(np.ascontiguousarray(ct)), _p(out))
        return out

    def bfv_multiply(self, a, b):
        L = a.shape[1]
        out = np.empty((3, L, self.N), dtype=np.uint64)
        lib().ho_bfv_multiply(self.h, L, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
        return out

    def batch_op(self, op, a, idx_a, b, idx_b, relin_key=None, threads=0):
        """a, b: [n, 2, L, N] slabs. Returns [n_results, size_out, L_out, N]."""
        L = a.shape[2]
        n = len(idx_a)
        shape = {OP_ADD: (n, 2, L, self.N), OP_MUL: (n, 3, L, self.N), OP_MUL_RELIN: (n, 2, L, self.N),
                 OP_MUL_RELIN_RESCALE: (n, 2, L - 1, self.N)}[op]
        out = np.empty(shape, dtype=np.uint64)
        ia = np.ascontiguousarray(idx_a, dtype=np.uint32)
        ib = np.ascontiguousarray(idx_b, dtype=np.uint32)
        rk = _p(relin_key) if relin_key is not None else None
        lib().ho_batch_op(self.h, op, L, n, _p(a), _p32(ia), _p(b), _p32(ib), rk, _p(out), threads)
        return out

    @staticmethod
    def _gk(galois_keys):
        """dict {galois element: key array} -> (count, element array, pointer array); the arrays must outlive the call"""
        galois_keys = galois_keys or {}
        elts = np.array(sorted(galois_keys), dtype=np.uint32)
        ptrs = (_u64p * max(1, len(elts)))(*[_p(galois_keys[int(e)]) for e in elts])
        return len(elts), elts, ptrs

    def rotate(self, ct, step, galois_keys):
        """Evaluator::rotate_vector / rotate_rows with SEAL's NAF decomposition over the given keys"""
        L = ct.shape[1]
        ct = np.ascontiguousarray(ct)
        out = np.empty_like(ct)
        n, elts, ptrs = self._gk(galois_keys)
        if lib().ho_rotate(self.h, L, step, n, _p32(elts) if n else None, ptrs, _p(ct), _p(out)):
            raise KeyError("Galois key not present / step count too large")
        return out

    def accumulate(self, ct, count, galois_keys):
        """SEALContextWrapper::accumulateCKKS / accumulateBFV (count > 0)"""
        L = ct.shape[1]
        out = np.ascontiguousarray(ct).copy()
        n, elts, ptrs = self._gk(galois_keys)
        if lib().ho_accumulate(self.h, L, count, n, _p32(elts) if n else None, ptrs, _p(out)):
            raise KeyError("Galois key not present / count 0")
        return out

    def batch_outer(self, op, a, b, relin_key=None, galois_keys=None, count=0, threads=0):
        """operate() of the element-wise / dot-product workloads: a [b0, 2, L, N], b [b1, 2, L, N] -> [b0*b1, size, L', N], result
        i*b1 + x = op(a[i], b[x]); OpenMP `parallel for collapse(2)` as the reference (ckks eltwise .cpp:325)."""
        b0, b1, L = a.shape[0], b.shape[0], a.shape[2]
        n = b0 * b1
        shape = {OP_ADD: (n, 2, L, self.N), OP_MUL: (n, 3, L, self.N), OP_MUL_RELIN: (n, 2, L, self.N),
                 OP_MUL_RELIN_RESCALE: (n, 2, L - 1, self.N), OP_DOT: (n, 2, L, self.N)}[op]
        out = np.empty(shape, dtype=np.uint64)
        ng, elts, ptrs = self._gk(galois_keys)
        rk = _p(relin_key) if relin_key is not None else None
        if lib().ho_batch_outer(self.h, op, L, b0, b1, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), rk, ng, _p32(elts) if ng else None, ptrs,
                                count, _p(out), threads):
            raise KeyError("Galois key not present")
        return out

    def bfv_matmul_rows(self, A, B, relin_key, galois_keys, dim2, threads=0):
        """MatMultRowLatencyBenchmark::matmultrow (bfv row .cpp:486-539): A [n, 2, L, N] row-pair ciphertexts, B [2, L, N]"""
        n, L = A.shape[0], A.shape[2]
        out = np.empty_like(A)
        ng, elts, ptrs = self._gk(galois_keys)
        if lib().ho_bfv_matmul_rows(self.h, L, n, _p(np.ascontiguousarray(A)), _p(np.ascontiguousarray(B)), _p(relin_key), ng, _p32(elts) if ng else None, ptrs,
                                    dim2, _p(out), threads):
            raise KeyError("Galois key not present")
        return out

    # -- keys / encryption -----------------------------------------------------------------------
    def keygen_secret(self, seed):
        sk = np.empty((self.K, self.N), dtype=np.uint64)
        lib().ho_keygen_secret(self.h, seed, _p(sk))
        return sk

    def keygen_public(self, sk, seed):
        pk = np.empty((2, self.K, self.N), dtype=np.uint64)
        lib().ho_keygen_public(self.h, _p(sk), seed, _p(pk))
        return pk

    def keygen_relin(self, sk, seed):
        out = np.empty((self.L, 2, self.K, self.N), dtype=np.uint64)
        lib().ho_keygen_relin(self.h, _p(sk), seed, _p(out))
        return out

    def keygen_galois(self, sk, elt, seed):
        out = np.empty((self.L, 2, self.K, self.N), dtype=np.uint64)
        lib().ho_keygen_galois(self.h, _p(sk), elt, seed, _p(out))
        return out

    def encrypt(self, pk, plain, seed):
        out = np.empty((2, self.L, self.N), dtype=np.uint64)
        lib().ho_encrypt(self.h, _p(pk), _p(np.ascontiguousarray(plain, dtype=np.uint64)), seed, _p(out))
        return out

    def encrypt_explicit(self, pk, plain, u_small, e0_small, e1_small):
        """ho_encrypt with the three sampled polynomials given (small signed coefficients, int32[N])."""
        out = np.empty((2, self.L, self.N), dtype=np.uint64)
        i32p = C.POINTER(C.c_int32)
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (u_small, e0_small, e1_small)]
        lib().ho_encrypt_explicit(self.h, _p(pk), _p(np.ascontiguousarray(plain, dtype=np.uint64)), *[x.ctypes.data_as(i32p) for x in a], _p(out))
        return out

    def decrypt_phase(self, ct, sk):
        size, L, _ = ct.shape
        out = np.empty((L, self.N), dtype=np.uint64)
        lib().ho_decrypt_phase(self.h, L, size, _p(np.ascontiguousarray(ct)), _p(sk), _p(out))
        return out

    def bfv_decode_phase(self, phase):
        L = phase.shape[0]
        out = np.empty(self.N, dtype=np.uint64)
        lib().ho_bfv_decode_phase(self.h, L, _p(np.ascontiguousarray(phase)), _p(out))
        return out

    def crt_to_double(self, coeff_poly, inv_scale):
        L = coeff_poly.shape[0]
        out = np.empty(self.N, dtype=np.float64)
        lib().ho_crt_to_double(self.h, L, _p(np.ascontiguousarray(coeff_poly)), inv_scale,
                               out.ctypes.data_as(C.POINTER(C.c_double)))
        return out


# ---- encoders in numpy (test-side only; slot <-> evaluation-point map as SEAL's encoders) ------------
def _slot_exponents(N: int) -> np.ndarray:
    """Odd exponents 3^i mod 2N for slot i (ckks.cpp / batchencoder.cpp matrix_reps_index_map)."""
    m = 2 * N
    e = np.empty(N // 2, dtype=np.int64)
    pos = 1
    for i in range(N // 2):
        e[i] = pos
        pos = (pos * 3) % m
    return e


def ckks_encode(ctx: Context, values, scale: float, L: int | None = None) -> np.ndarray:
    """values (<= N/2 complex/real) -> plaintext residues [L][N] in NTT form."""
    N = ctx.N
    L = ctx.L if L is None else L
    v = np.zeros(N // 2, dtype=np.complex128)
    v[: len(values)] = values
    e = _slot_exponents(N)
    z = np.zeros(N, dtype=np.complex128)           # z[j] = p(zeta^(2j+1))
    z[(e - 1) // 2] = v
    z[(2 * N - e - 1) // 2] = np.conj(v)
    n = np.arange(N)
    coeffs = np.fft.fft(z) / N * np.exp(-1j * np.pi * n / N)
    c = np.rint(coeffs.real * scale)
    out = np.empty((L, N), dtype=np.uint64)
    ci = [int(x) for x in c]
    for i in range(L):
        q = ctx.moduli[i]
        out[i] = ctx.ntt(i, np.array([x % q for x in ci], dtype=np.uint64))
    return out


def ckks_decode(ctx: Context, plain_ntt: np.ndarray, scale: float) -> np.ndarray:
    """plaintext residues [L][N] NTT form -> N/2 complex slots."""
    N = ctx.N
    L = plain_ntt.shape[0]
    coeff = np.stack([ctx.intt(i, plain_ntt[i]) for i in range(L)])
    c = ctx.crt_to_double(coeff, 1.0 / scale)
    n = np.arange(N)
    z = N * np.fft.ifft(c * np.exp(1j * np.pi * n / N))
    e = _slot_exponents(N)
    return z[(e - 1) // 2]


class BatchCodec:
    """BFV BatchEncoder restated with a one-prime helper context mod t."""

    def __init__(self, N: int, t: int):
        self.N, self.t = N, t
        self.tctx = Context(SCHEME_CKKS, N, primes=[t])
        logn = N.bit_length() - 1
        e = _slot_exponents(N)

        def br(x):
            return int(format(x, f"0{logn}b")[::-1], 2)
        self.idx = np.array([br((int(x) - 1) // 2) for x in e] + [br((2 * N - int(x) - 1) // 2) for x in e])

    def encode(self, values) -> np.ndarray:
        v = np.zeros(self.N, dtype=np.int64)
        v[: len(values)] = values
        buf = np.zeros(self.N, dtype=np.uint64)
        buf[self.idx] = np.mod(v, self.t).astype(np.uint64)
        return self.tctx.intt(0, buf)

    def decode(self, plain) -> np.ndarray:
        ev = self.tctx.ntt(0, plain)
        return ev[self.idx].astype(np.int64)
