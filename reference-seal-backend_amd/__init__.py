"""MI355X-native HEBench backend — Python loader for the C ABI (include/he355.h).

The product is the shared library ``lib/libhebench_mi355x_backend.so`` (HIP kernels + C++17 host + the
HEBench API-Bridge entry points).  This module only binds its C ABI with ctypes so that tests, ``bench.py``
and ``__graft_entry__`` can drive it; it contains no arithmetic and never imports ``oracle``.

Import with ``importlib.import_module("reference-seal-backend_amd")`` (the directory name has a hyphen).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HE355_LIB_PATH selects another build of the same library (A/B timing of kernel variants, sanitizer builds): the product
# file is never swapped in place
LIB_PATH = os.environ.get("HE355_LIB_PATH") or os.path.join(_HERE, "lib", "libhebench_mi355x_backend.so")

SCHEME_BFV, SCHEME_CKKS = 1, 2
OK, E_INVALID_ARGS, E_PARAMS, E_DEVICE = 0, 1, 2, 3


class HE355Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"he355 error {code}: {msg}")
        self.code = code


class Indexer(C.Structure):
    _fields_ = [("a_base", C.c_uint64), ("b_base", C.c_uint64), ("b1", C.c_uint64),
                ("pairwise", C.c_int32), ("reserved", C.c_int32)]


class AllocStats(C.Structure):
    _fields_ = [("raw_mallocs", C.c_uint64), ("raw_frees", C.c_uint64), ("pool_hits", C.c_uint64), ("pool_misses", C.c_uint64),
                ("cached_bytes", C.c_uint64), ("live_bytes", C.c_uint64)]


class PathStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("ks_fused", "ks_unfused", "ks_latency", "ks_lds", "level_sums_in_k3", "level_sums_by_kernel",
                                           "level_sum_launches_in_k3", "reserved")]


def build(force: bool = False) -> str:
    """Compile the backend for gfx950 with hipcc (csrc/Makefile)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.run(["make", "-C", csrc, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", csrc, "-j8", "-s"], check=True)
    return LIB_PATH


_lib = None
_u64p = C.POINTER(C.c_uint64)


def lib():
    """The loaded shared library.  Fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no fallback path)")
        L = C.CDLL(LIB_PATH)
        vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
        vpp = C.POINTER(vp)
        u8p = C.POINTER(C.c_uint8)
        sig = {
            "he355_last_error": (C.c_char_p, []),
            "he355_ctx_create": (i32, [i32, u64, C.POINTER(C.c_int32), u64, i32, i32, vpp]),
            "he355_ctx_create_primes": (i32, [i32, u64, _u64p, u64, u64, vpp]),
            "he355_ctx_destroy": (None, [vp]),
            "he355_poly_degree": (u64, [vp]), "he355_key_modulus_count": (u64, [vp]),
            "he355_data_modulus_count": (u64, [vp]), "he355_modulus": (u64, [vp, u64]),
            "he355_plain_modulus": (u64, [vp]), "he355_prime_uses_fp64": (i32, [vp, u64]),
            "he355_bfv_aux_base": (u64, [vp, i32, _u64p, u64]),
            "he355_galois_elt_from_step": (u32, [vp, i32]),
            "he355_galois_elts_all": (u64, [vp, C.POINTER(u32), u64]),
            "he355_device_count": (i32, [C.POINTER(i32)]),
            "he355_device_init": (i32, [vp, i32]),
            "he355_malloc": (i32, [vp, u64, vpp]), "he355_free": (i32, [vp, vp]),
            "he355_upload": (i32, [vp, vp, vp, u64]), "he355_download": (i32, [vp, vp, vp, u64]),
            "he355_copy": (i32, [vp, vp, vp, u64]),
            "he355_copy_peer": (i32, [vp, vp, vp, vp, u64]),
            "he355_sync": (i32, [vp]),
            "he355_fill_uniform": (i32, [vp, vp, u64, u8p, u32, u64]),
            "he355_fill_uniform_at": (i32, [vp, vp, u64, u8p, u32, u64, u64]),
            "he355_set_dual_stream": (i32, [vp, i32]),
            "he355_set_relin_key": (i32, [vp, _u64p]), "he355_set_galois_key": (i32, [vp, u32, _u64p]),
            "he355_set_relin_key_synthetic": (i32, [vp, u64]),
            "he355_set_galois_key_synthetic": (i32, [vp, u32, u64]),
            "he355_add": (i32, [vp, i32, i32, u64, vp, vp, Indexer, vp]),
            "he355_sub": (i32, [vp, i32, i32, u64, vp, vp, Indexer, vp]),
            "he355_multiply": (i32, [vp, i32, u64, vp, vp, Indexer, vp]),
            "he355_bfv_multiply": (i32, [vp, i32, u64, vp, vp, Indexer, vp]),
            "he355_multiply_relin": (i32, [vp, i32, u64, vp, vp, Indexer, i32, vp]),
            "he355_relinearize": (i32, [vp, i32, u64, vp, vp]),
            "he355_relinearize_rescale": (i32, [vp, i32, u64, vp, vp]),
            "he355_set_public_key": (i32, [vp, vp]), "he355_set_secret_key": (i32, [vp, vp]),
            "he355_encrypt": (i32, [vp, u64, vp, u64, u64, vp]),
            "he355_keygen_relin": (i32, [vp, u64]), "he355_keygen_galois": (i32, [vp, C.c_uint32, u64]),
            "he355_ckks_encode": (i32, [vp, u64, vp, u64, C.c_double, vp]),
            "he355_ckks_decode": (i32, [vp, i32, u64, vp, C.c_double, vp]),
            "he355_bfv_encode": (i32, [vp, u64, vp, u64, vp]),
            "he355_bfv_decode": (i32, [vp, u64, vp, vp]),
            "he355_ckks_decode_slots": (i32, [vp, i32, u64, vp, C.c_double, _u64p, u64, vp]),
            "he355_bfv_decode_slots": (i32, [vp, u64, vp, _u64p, u64, vp]),
            "he355_host_alloc": (i32, [vp, u64, vpp]), "he355_host_free": (i32, [vp, vp]),
            "he355_decrypt": (i32, [vp, i32, i32, u64, vp, vp]),
            "he355_multiply_plain": (i32, [vp, i32, i32, u64, vp, vp, Indexer, vp]),
            "he355_add_plain": (i32, [vp, i32, i32, u64, vp, vp, Indexer, vp]),
            "he355_mod_switch_drop": (i32, [vp, i32, i32, u64, vp, vp]),
            "he355_sum": (i32, [vp, i32, i32, u64, vp, vp]),
            "he355_multiply_accumulate": (i32, [vp, i32, u64, u64, u64, vp, u64, u64, vp, u64, u64, vp]),
            "he355_bfv_multiply_relin_accumulate": (i32, [vp, i32, u64, u64, u64, vp, u64, u64, vp, u64, u64, vp]),
            "he355_rescale": (i32, [vp, i32, i32, u64, vp, vp]),
            "he355_apply_galois": (i32, [vp, i32, u64, vp, u32, vp]),
            "he355_rotate": (i32, [vp, i32, u64, vp, i32, vp]),
            "he355_rotate_add": (i32, [vp, i32, u64, vp, i32, vp, vp]),
            "he355_rotate_each": (i32, [vp, i32, u64, vp, vp, vp]),
            "he355_rotate_sum": (i32, [vp, i32, u64, vp, vp, u64, vp, vp]),
            "he355_accumulate": (i32, [vp, i32, u64, vp, u64, vp]),
            "he355_encrypt_zero": (i32, [vp, u64, u64, u64, vp]),
            "he355_set_zero_stream": (i32, [vp, u64, u64]),
            "he355_ntt_forward": (i32, [vp, vp, u64, u8p, u32]),
            "he355_ntt_inverse": (i32, [vp, vp, u64, u8p, u32]),
            "he355_timer_begin": (i32, [vp]), "he355_timer_end": (i32, [vp, C.POINTER(C.c_float)]),
            "he355_probe_dominant_kernel": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "he355_clock_probe_begin": (i32, [vp, u64]), "he355_clock_probe_end": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
            "he355_set_chunk": (i32, [vp, u64]),
            "he355_set_latency_max": (i32, [vp, u64]),
            "he355_set_level_walk": (i32, [vp, i32]),
            "he355_set_lds_max": (i32, [vp, u64]),
            "he355_mem_info": (i32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "he355_alloc_stats": (i32, [vp, C.POINTER(AllocStats)]),
            "he355_pool_trim": (i32, [vp, C.POINTER(C.c_uint64)]),
            "he355_path_stats": (i32, [vp, C.POINTER(PathStats), i32]),
            "he355_bridge_abi": (u64, [C.c_char_p, u64]),
            "he355_bridge_group_load_bytes": (u64, [i32, i32]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


C_ABI_SYMBOLS = [
    "he355_last_error", "he355_ctx_create", "he355_ctx_create_primes", "he355_ctx_destroy", "he355_poly_degree",
    "he355_key_modulus_count", "he355_data_modulus_count", "he355_modulus", "he355_plain_modulus",
    "he355_prime_uses_fp64", "he355_bfv_aux_base", "he355_galois_elt_from_step", "he355_galois_elts_all", "he355_device_count",
    "he355_device_init", "he355_malloc", "he355_free", "he355_upload", "he355_download", "he355_copy", "he355_copy_peer", "he355_sync",
    "he355_fill_uniform", "he355_fill_uniform_at", "he355_set_dual_stream", "he355_set_relin_key", "he355_set_galois_key", "he355_set_relin_key_synthetic",
    "he355_set_galois_key_synthetic", "he355_add", "he355_sub", "he355_multiply", "he355_bfv_multiply", "he355_multiply_relin",
    "he355_relinearize", "he355_relinearize_rescale", "he355_multiply_accumulate", "he355_bfv_multiply_relin_accumulate", "he355_multiply_plain", "he355_add_plain",
    "he355_mod_switch_drop", "he355_sum", "he355_set_public_key", "he355_set_secret_key", "he355_encrypt", "he355_decrypt", "he355_keygen_relin", "he355_keygen_galois", "he355_ckks_encode", "he355_ckks_decode",
    "he355_bfv_encode", "he355_bfv_decode", "he355_ckks_decode_slots", "he355_bfv_decode_slots", "he355_host_alloc", "he355_host_free", "he355_rescale", "he355_apply_galois", "he355_rotate", "he355_rotate_add", "he355_rotate_each", "he355_rotate_sum", "he355_accumulate", "he355_encrypt_zero", "he355_set_zero_stream",
    "he355_ntt_forward", "he355_ntt_inverse", "he355_timer_begin", "he355_timer_end", "he355_probe_dominant_kernel", "he355_clock_probe_begin", "he355_clock_probe_end", "he355_set_chunk", "he355_set_latency_max", "he355_set_level_walk", "he355_set_lds_max", "he355_mem_info", "he355_alloc_stats", "he355_pool_trim", "he355_path_stats", "he355_bridge_abi", "he355_bridge_group_load_bytes",
]


def _check(code: int):
    if code != 0:
        raise HE355Error(code, lib().he355_last_error().decode())


def chain_bits(depth: int, coeff_bits: int) -> list[int]:
    """{60, bits x (depth-1), 60}: the reference's rule (seal_context.cpp:79-82,107-110)."""
    return [60] + [coeff_bits] * (depth - 1) + [60]


class DeviceBuffer:
    """A slab of uint64 in HBM owned by a Context."""

    def __init__(self, ctx: "Context", n_u64: int):
        self.ctx, self.n = ctx, int(n_u64)
        p = C.c_void_p()
        _check(lib().he355_malloc(ctx.h, self.n * 8, C.byref(p)))
        self.ptr = p

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        assert a.size == self.n, (a.size, self.n)
        _check(lib().he355_upload(self.ctx.h, self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return self

    def download(self, shape=None) -> np.ndarray:
        out = np.empty(self.n, dtype=np.uint64)
        _check(lib().he355_download(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out.reshape(shape) if shape is not None else out

    def download_range(self, offset_u64: int, shape) -> np.ndarray:
        """prod(shape) elements starting at element offset_u64 (one row out of the middle or the end of a large slab)"""
        k = int(np.prod(shape))
        assert 0 <= offset_u64 and offset_u64 + k <= self.n, (offset_u64, k, self.n)
        out = np.empty(k, dtype=np.uint64)
        if k:
            _check(lib().he355_download(self.ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr.value + int(offset_u64) * 8), out.nbytes))
        return out.reshape(shape)

    def download_head(self, shape) -> np.ndarray:
        """the first prod(shape) elements only (a sample of a large slab without moving the whole slab over PCIe)"""
        k = int(np.prod(shape))
        assert 0 <= k <= self.n, (k, self.n)
        out = np.empty(k, dtype=np.uint64)
        if k:
            _check(lib().he355_download(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out.reshape(shape)

    def copy_into(self, dst: "DeviceBuffer", dst_offset_u64: int = 0, n_u64: int | None = None):
        """device-to-device copy of the first n_u64 words of this slab to dst[dst_offset_u64:] (on the context's stream)"""
        n = self.n if n_u64 is None else int(n_u64)
        assert n <= self.n and dst_offset_u64 + n <= dst.n
        _check(lib().he355_copy(self.ctx.h, C.c_void_p(dst.ptr.value + dst_offset_u64 * 8), self.ptr, n * 8))

    def free(self):
        if self.ptr:
            lib().he355_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """Host mirror of the reference's SEALContextWrapper for the hot path: parameters + device state."""

    def __init__(self, scheme: int, N: int, bit_sizes=None, primes=None, plain_bits: int = 0,
                 plain_modulus: int = 0, sec128: bool = True, device: int | None = None):
        L = lib()
        h = C.c_void_p()
        if primes is not None:
            arr = np.asarray(primes, dtype=np.uint64)
            _check(L.he355_ctx_create_primes(scheme, N, arr.ctypes.data_as(_u64p), len(arr), plain_modulus, C.byref(h)))
        else:
            bs = (C.c_int32 * len(bit_sizes))(*bit_sizes)
            _check(L.he355_ctx_create(scheme, N, bs, len(bit_sizes), plain_bits, int(sec128), C.byref(h)))
        self.h = h
        self.scheme, self.N = scheme, N
        self.K = L.he355_key_modulus_count(h)
        self.L = L.he355_data_modulus_count(h)
        self.moduli = [L.he355_modulus(h, i) for i in range(self.K)]
        self.fp64 = [bool(L.he355_prime_uses_fp64(h, i)) for i in range(self.K)]
        self.t = L.he355_plain_modulus(h)
        self._bufs = []
        if device is not None:
            self.device_init(device)

    def bfv_aux_base(self, level: int | None = None) -> list[int]:
        """[m_sk, B_0, B_1, ...]: the auxiliary base of the BEHZ multiply at `level` data primes (host-side; no device needed)."""
        out = np.zeros(64, dtype=np.uint64)
        n = lib().he355_bfv_aux_base(self.h, self.L if level is None else level, out.ctypes.data_as(_u64p), 64)
        return [int(v) for v in out[:n]]

    def device_init(self, device: int = 0):
        _check(lib().he355_device_init(self.h, device))

    def close(self):
        if self.h:
            for b in self._bufs:
                b.free()
            self._bufs = []
            lib().he355_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ----------------------------------------------------------------------------------
    def alloc(self, n_u64: int) -> DeviceBuffer:
        b = DeviceBuffer(self, n_u64)
        self._bufs.append(b)
        return b

    def to_device(self, arr: np.ndarray) -> DeviceBuffer:
        return self.alloc(arr.size).upload(arr)

    def fill_uniform(self, buf: DeviceBuffer, n_polys: int, prime_of, seed: int, first_poly: int = 0):
        pm = (C.c_uint8 * len(prime_of))(*prime_of)
        _check(lib().he355_fill_uniform_at(self.h, buf.ptr, n_polys, pm, len(prime_of), seed, first_poly))

    def set_dual_stream(self, on: bool):
        _check(lib().he355_set_dual_stream(self.h, int(on)))

    def sync(self):
        _check(lib().he355_sync(self.h))

    def set_chunk(self, ops: int):
        _check(lib().he355_set_chunk(self.h, ops))

    # -- keys ------------------------------------------------------------------------------------
    def set_relin_key(self, key: np.ndarray):
        k = np.ascontiguousarray(key, dtype=np.uint64)
        assert k.size == self.L * 2 * self.K * self.N
        _check(lib().he355_set_relin_key(self.h, k.ctypes.data_as(_u64p)))

    def set_galois_key(self, elt: int, key: np.ndarray):
        k = np.ascontiguousarray(key, dtype=np.uint64)
        assert k.size == self.L * 2 * self.K * self.N
        _check(lib().he355_set_galois_key(self.h, elt, k.ctypes.data_as(_u64p)))

    def keygen_relin(self, seed: int):
        _check(lib().he355_keygen_relin(self.h, seed))

    def keygen_galois(self, elt: int, seed: int):
        _check(lib().he355_keygen_galois(self.h, elt, seed))

    def set_relin_key_synthetic(self, seed: int):
        _check(lib().he355_set_relin_key_synthetic(self.h, seed))

    def set_galois_key_synthetic(self, elt: int, seed: int):
        _check(lib().he355_set_galois_key_synthetic(self.h, elt, seed))

    def galois_elt(self, step: int) -> int:
        return lib().he355_galois_elt_from_step(self.h, step)

    def galois_elts_all(self):
        buf = (C.c_uint32 * 64)()
        n = lib().he355_galois_elts_all(self.h, buf, 64)
        return [int(x) for x in buf[:n]]

    # -- evaluator (device slabs) ----------------------------------------------------------------
    @staticmethod
    def outer(a_base: int, b0: int, b_base: int, b1: int) -> Indexer:
        """HEBench indexers: result r = i*b1 + x uses operand0[a_base+i], operand1[b_base+x]."""
        return Indexer(a_base, b_base, b1, 0, 0)

    @staticmethod
    def pairwise(a_base: int = 0, b_base: int = 0) -> Indexer:
        return Indexer(a_base, b_base, 1, 1, 0)

    def add(self, L, size, n, a, b, ix, out, sub=False):
        f = lib().he355_sub if sub else lib().he355_add
        _check(f(self.h, L, size, n, a.ptr, b.ptr, ix, out.ptr))

    def multiply(self, L, n, a, b, ix, out):
        _check(lib().he355_multiply(self.h, L, n, a.ptr, b.ptr, ix, out.ptr))

    def bfv_multiply(self, L, n, a, b, ix, out):
        _check(lib().he355_bfv_multiply(self.h, L, n, a.ptr, b.ptr, ix, out.ptr))

    def multiply_relin(self, L, n, a, b, ix, out, rescale=False):
        _check(lib().he355_multiply_relin(self.h, L, n, a.ptr, b.ptr, ix, int(rescale), out.ptr))

    def relinearize(self, L, n, ct3, out):
        _check(lib().he355_relinearize(self.h, L, n, ct3.ptr, out.ptr))

    def multiply_plain(self, L, size, n, ct, pt, ix, out):
        _check(lib().he355_multiply_plain(self.h, L, size, n, ct.ptr, pt.ptr, ix, out.ptr))

    def add_plain(self, L, size, n, ct, pt, ix, out):
        _check(lib().he355_add_plain(self.h, L, size, n, ct.ptr, pt.ptr, ix, out.ptr))

    def mod_switch_drop(self, L, L_to, n_polys, inp, out):
        _check(lib().he355_mod_switch_drop(self.h, L, L_to, n_polys, inp.ptr, out.ptr))

    def sum(self, L, size, n, inp, out):
        _check(lib().he355_sum(self.h, L, size, n, inp.ptr, out.ptr))

    def set_public_key(self, pk: np.ndarray):
        pk = np.ascontiguousarray(pk, dtype=np.uint64)
        _check(lib().he355_set_public_key(self.h, pk.ctypes.data))

    def set_secret_key(self, sk: np.ndarray):
        sk = np.ascontiguousarray(sk, dtype=np.uint64)
        _check(lib().he355_set_secret_key(self.h, sk.ctypes.data))

    def encrypt(self, n, plain, seed, first_index, out):
        _check(lib().he355_encrypt(self.h, n, plain.ptr, seed, first_index, out.ptr))

    def encrypt_zero(self, n, seed, first_index, out):
        _check(lib().he355_encrypt_zero(self.h, n, seed, first_index, out.ptr))

    def set_zero_stream(self, seed, first_index=0):
        _check(lib().he355_set_zero_stream(self.h, seed, first_index))

    def decrypt(self, L, size, n, ct, out):
        _check(lib().he355_decrypt(self.h, L, size, n, ct.ptr, out.ptr))

    def ckks_encode(self, n, values, count, scale, plain):
        _check(lib().he355_ckks_encode(self.h, n, values.ptr, count, scale, plain.ptr))

    def ckks_decode(self, L, n, plain, scale, out):
        _check(lib().he355_ckks_decode(self.h, L, n, plain.ptr, scale, out.ptr))

    def bfv_encode(self, n, values, count, plain):
        _check(lib().he355_bfv_encode(self.h, n, values.ptr, count, plain.ptr))

    def bfv_decode(self, n, plain, out):
        _check(lib().he355_bfv_decode(self.h, n, plain.ptr, out.ptr))

    @staticmethod
    def _ranges(ranges):
        flat = [int(v) for r in ranges for v in r]
        return (C.c_uint64 * len(flat))(*flat), len(flat) // 2

    def ckks_decode_slots(self, L, n, plain, scale, ranges, out):
        """only the slots of `ranges` = [(first, count), ...]: out is [n][sum of counts]"""
        arr, k = self._ranges(ranges)
        _check(lib().he355_ckks_decode_slots(self.h, L, n, plain.ptr, scale, arr, k, out.ptr))

    def bfv_decode_slots(self, n, plain, ranges, out):
        arr, k = self._ranges(ranges)
        _check(lib().he355_bfv_decode_slots(self.h, n, plain.ptr, arr, k, out.ptr))

    def relinearize_rescale(self, L, n, ct3, out):
        _check(lib().he355_relinearize_rescale(self.h, L, n, ct3.ptr, out.ptr))

    def multiply_accumulate(self, L, rows, cols, inner, a, a_stride_i, a_stride_k, b, b_stride_k, b_stride_j, out):
        _check(lib().he355_multiply_accumulate(self.h, L, rows, cols, inner, a.ptr, a_stride_i, a_stride_k, b.ptr, b_stride_k, b_stride_j, out.ptr))

    def bfv_multiply_relin_accumulate(self, L, rows, cols, inner, a, a_stride_i, a_stride_k, b, b_stride_k, b_stride_j, out):
        _check(lib().he355_bfv_multiply_relin_accumulate(self.h, L, rows, cols, inner, a.ptr, a_stride_i, a_stride_k, b.ptr, b_stride_k,
                                                         b_stride_j, out.ptr))

    def rescale(self, L, size, n, inp, out):
        _check(lib().he355_rescale(self.h, L, size, n, inp.ptr, out.ptr))

    def apply_galois(self, L, n, inp, elt, out):
        _check(lib().he355_apply_galois(self.h, L, n, inp.ptr, elt, out.ptr))

    def rotate(self, L, n, inp, step, out):
        _check(lib().he355_rotate(self.h, L, n, inp.ptr, step, out.ptr))

    def rotate_each(self, L, n, inp, steps, out):
        arr = (C.c_int32 * n)(*[int(v) for v in steps])
        _check(lib().he355_rotate_each(self.h, L, n, inp.ptr, arr, out.ptr))

    def set_latency_max(self, n=None):
        """largest batch whose key switches take the latency shape; None: the library's rule (batch * N <= 2^17, at most 12)"""
        _check(lib().he355_set_latency_max(self.h, 2 ** 64 - 1 if n is None else n))

    def set_lds_max(self, n=None):
        """largest batch whose key switches take the ring-in-LDS shape (N <= 8192); None: the library's rule; 0: never"""
        _check(lib().he355_set_lds_max(self.h, 2 ** 64 - 1 if n is None else n))

    def set_level_walk(self, on: bool):
        _check(lib().he355_set_level_walk(self.h, int(on)))

    def mem_info(self):
        """(free, total) bytes of the context's device"""
        f, t = C.c_uint64(), C.c_uint64()
        _check(lib().he355_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def alloc_stats(self) -> dict:
        """raw hipMalloc / hipFree calls and pool hits / misses of this context since device_init"""
        st = AllocStats()
        _check(lib().he355_alloc_stats(self.h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in AllocStats._fields_}

    def path_stats(self, reset: bool = False) -> dict:
        """which key-switch shape / schedule ran (he355_path_stats): counts of kernel sequences since device_init or the last reset"""
        st = PathStats()
        _check(lib().he355_path_stats(self.h, C.byref(st), int(reset)))
        return {k: int(getattr(st, k)) for k, _ in PathStats._fields_ if k != "reserved"}

    def pool_trim(self) -> int:
        b = C.c_uint64()
        _check(lib().he355_pool_trim(self.h, C.byref(b)))
        return int(b.value)

    def rotate_sum(self, L, n, inp, steps, out):
        """out = inp + sum_j rotate(inp, steps[j]) with shared NAF prefixes; returns the Galois key switches issued per ciphertext"""
        arr = (C.c_int32 * len(steps))(*[int(v) for v in steps])
        ks = C.c_uint64(0)
        _check(lib().he355_rotate_sum(self.h, L, n, inp.ptr, arr, len(steps), out.ptr, C.byref(ks)))
        return int(ks.value)

    def rotate_add(self, L, n, inp, step, addend, out):
        _check(lib().he355_rotate_add(self.h, L, n, inp.ptr, step, addend.ptr, out.ptr))

    def accumulate(self, L, n, inout, count, tmp):
        _check(lib().he355_accumulate(self.h, L, n, inout.ptr, count, tmp.ptr))

    def ntt(self, buf, n_polys, prime_of, inverse=False):
        pm = (C.c_uint8 * len(prime_of))(*prime_of)
        f = lib().he355_ntt_inverse if inverse else lib().he355_ntt_forward
        _check(f(self.h, buf.ptr, n_polys, pm, len(prime_of)))

    # -- timing on the kernels' stream -----------------------------------------------------------
    def timer_begin(self):
        _check(lib().he355_timer_begin(self.h))

    def probe_dominant_kernel(self):
        """(total ms, launches, ops) of the k_k3<fp64> launches inside the last timer_begin/timer_end region."""
        ms, n, ops = C.c_float(), C.c_uint64(), C.c_uint64()
        _check(lib().he355_probe_dominant_kernel(self.h, C.byref(ms), C.byref(n), C.byref(ops)))
        return float(ms.value), int(n.value), int(ops.value)

    def clock_probe_begin(self, duration_us: int):
        """one wave on its own stream samples shader cycles against the 100 MHz counter for `duration_us` while what is queued next runs"""
        _check(lib().he355_clock_probe_begin(self.h, int(duration_us)))

    def clock_probe_end(self):
        """(MHz held, seconds covered) of the probe started by clock_probe_begin"""
        mhz, sec = C.c_double(0), C.c_double(0)
        _check(lib().he355_clock_probe_end(self.h, C.byref(mhz), C.byref(sec)))
        return mhz.value, sec.value

    def timer_end(self) -> float:
        ms = C.c_float()
        _check(lib().he355_timer_end(self.h, C.byref(ms)))
        return float(ms.value)


def process_alloc_stats() -> dict:
    """totals over every context of this process (the benchmark objects behind API-Bridge handles own theirs)"""
    st = AllocStats()
    _check(lib().he355_alloc_stats(None, C.byref(st)))
    return {k: int(getattr(st, k)) for k, _ in AllocStats._fields_}


def device_count() -> int:
    n = C.c_int(0)
    lib().he355_device_count(C.byref(n))
    return int(n.value)
