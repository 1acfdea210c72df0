"""Mini test harness: drives libhebench_mi355x_backend.so through the HEBench API-Bridge C ABI the way
test_harness does (initEngine -> subscribe -> describe -> createBenchmark -> initBenchmark -> encode -> encrypt ->
load -> operate -> store -> decrypt -> decode), with the struct layouts of include/hebench_api_bridge.h."""
from __future__ import annotations

import ctypes as C

import numpy as np

MAX_BUF, MAX_OP = 256, 32
ECODE_INVALID_ARGS, ECODE_CRITICAL = 0x7FFFFFFE, 0x7FFFFFFF
# enumerators of include/hebench_api_bridge.h ([UPSTREAM-UNVERIFIED] recollection of api-bridge v0.8: Workload and DataType
# are numbered from 1).  tests/test_api_bridge_cpu.py::test_harness_constants_match_the_library holds every number in this
# file to what the library was really compiled with (he355_bridge_abi).
LATENCY, OFFLINE = 0, 1
W_MATMUL, W_MUL, W_ADD, W_DOT = 1, 2, 3, 4
W_LOGREG3 = 6  # LogisticRegression_PolyD3
DT_INT64, DT_FLOAT64 = 2, 4
WP_UINT64 = 1
SCHEME_CKKS, SCHEME_BFV = 100, 101


class Handle(C.Structure):
    _fields_ = [("p", C.c_void_p), ("size", C.c_uint64), ("tag", C.c_int64)]


NativeDataBuffer = Handle


class DataPack(C.Structure):
    _fields_ = [("p_buffers", C.POINTER(NativeDataBuffer)), ("buffer_count", C.c_uint64), ("param_position", C.c_uint64)]


class DataPackCollection(C.Structure):
    _fields_ = [("p_data_packs", C.POINTER(DataPack)), ("pack_count", C.c_uint64)]


class ParameterIndexer(C.Structure):
    _fields_ = [("value_index", C.c_uint64), ("batch_size", C.c_uint64)]


class _Lat(C.Structure):
    _fields_ = [("warmup_iterations_count", C.c_uint64)]


class _Off(C.Structure):
    _fields_ = [("data_count", C.c_uint64 * MAX_OP)]


class _CatU(C.Union):
    _fields_ = [("reserved", C.c_uint64 * (2 * MAX_OP)), ("latency", _Lat), ("offline", _Off)]


class CategoryParams(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("min_test_time_ms", C.c_uint64), ("u", _CatU)]


class BenchmarkDescriptor(C.Structure):
    _fields_ = [("workload", C.c_int), ("data_type", C.c_int), ("category", C.c_int), ("cat_params", CategoryParams),
                ("cipher_param_mask", C.c_uint32), ("scheme", C.c_int32), ("security", C.c_int32), ("other", C.c_int64)]


class _WPU(C.Union):
    _fields_ = [("i_param", C.c_int64), ("u_param", C.c_uint64), ("f_param", C.c_double)]


class WorkloadParam(C.Structure):
    _anonymous_ = ("v",)
    _fields_ = [("data_type", C.c_int), ("name", C.c_char * MAX_BUF), ("v", _WPU)]


class WorkloadParams(C.Structure):
    _fields_ = [("params", C.POINTER(WorkloadParam)), ("count", C.c_uint64)]


class BridgeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"API bridge error {code:#x}: {msg}")
        self.code = code


class Backend:
    def __init__(self, lib_path: str):
        L = self.L = C.CDLL(lib_path)
        hp = C.POINTER(Handle)
        L.initEngine.argtypes = [hp, C.c_void_p, C.c_uint64]
        L.destroyHandle.argtypes = [Handle]
        L.subscribeBenchmarksCount.argtypes = [Handle, C.POINTER(C.c_uint64)]
        L.subscribeBenchmarks.argtypes = [Handle, hp, C.c_uint64]
        L.getWorkloadParamsDetails.argtypes = [Handle, Handle, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.describeBenchmark.argtypes = [Handle, Handle, C.POINTER(BenchmarkDescriptor), C.POINTER(WorkloadParams), C.c_uint64]
        L.createBenchmark.argtypes = [Handle, Handle, C.POINTER(WorkloadParams), hp]
        L.initBenchmark.argtypes = [Handle, C.POINTER(BenchmarkDescriptor)]
        L.encode.argtypes = [Handle, C.POINTER(DataPackCollection), hp]
        L.decode.argtypes = [Handle, Handle, C.POINTER(DataPackCollection)]
        L.encrypt.argtypes = [Handle, Handle, hp]
        L.decrypt.argtypes = [Handle, Handle, hp]
        L.load.argtypes = [Handle, hp, C.c_uint64, hp]
        L.store.argtypes = [Handle, Handle, hp, C.c_uint64]
        L.operate.argtypes = [Handle, Handle, C.POINTER(ParameterIndexer), C.c_uint64, hp]
        for f in ("getSchemeName", "getSchemeSecurityName", "getBenchmarkDescriptionEx", "getErrorDescription", "getLastErrorDescription"):
            getattr(L, f).restype = C.c_uint64
        L.getSchemeName.argtypes = [Handle, C.c_int32, C.c_char_p, C.c_uint64]
        L.getSchemeSecurityName.argtypes = [Handle, C.c_int32, C.c_int32, C.c_char_p, C.c_uint64]
        L.getBenchmarkDescriptionEx.argtypes = [Handle, Handle, C.POINTER(WorkloadParams), C.c_char_p, C.c_uint64]
        L.getErrorDescription.argtypes = [Handle, C.c_int32, C.c_char_p, C.c_uint64]
        L.getLastErrorDescription.argtypes = [Handle, C.c_char_p, C.c_uint64]
        self.engine = Handle()
        self.chk(L.initEngine(C.byref(self.engine), None, 0))

    def last_error(self) -> str:
        n = self.L.getLastErrorDescription(self.engine, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        self.L.getLastErrorDescription(self.engine, buf, n + 1)
        return buf.value.decode()

    def chk(self, code):
        if code != 0:
            raise BridgeError(code & 0xFFFFFFFF if code < 0 else code, self.last_error())

    def _str(self, fn, *args):
        n = fn(*args, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        fn(*args, buf, n + 1)
        return buf.value.decode()

    def scheme_name(self, s):
        return self._str(self.L.getSchemeName, self.engine, s)

    def security_name(self, s, sec):
        return self._str(self.L.getSchemeSecurityName, self.engine, s, sec)

    def error_description(self, code):
        return self._str(self.L.getErrorDescription, self.engine, code)

    def benchmarks(self):
        n = C.c_uint64()
        self.chk(self.L.subscribeBenchmarksCount(self.engine, C.byref(n)))
        hs = (Handle * n.value)()
        self.chk(self.L.subscribeBenchmarks(self.engine, hs, n.value))
        out = []
        for h in hs:
            pc, dc = C.c_uint64(), C.c_uint64()
            self.chk(self.L.getWorkloadParamsDetails(self.engine, h, C.byref(pc), C.byref(dc)))
            desc = BenchmarkDescriptor()
            sets = [(WorkloadParam * pc.value)() for _ in range(dc.value)]
            wps = (WorkloadParams * max(1, dc.value))()
            for i, s in enumerate(sets):
                wps[i].params = s
                wps[i].count = pc.value
            self.chk(self.L.describeBenchmark(self.engine, h, C.byref(desc), wps, dc.value))
            defaults = [[(p.name.decode(), p.u_param) for p in s] for s in sets]
            out.append(dict(handle=Handle(h.p, h.size, h.tag), desc=desc, defaults=defaults))
        return out

    def find(self, workload, scheme, category):
        for b in self.benchmarks():
            d = b["desc"]
            if d.workload == workload and d.scheme == scheme and d.category == category:
                return b
        raise KeyError((workload, scheme, category))

    def description_text(self, bench, params):
        wp, _keep = self._wparams(params)
        return self._str(self.L.getBenchmarkDescriptionEx, self.engine, bench["handle"], C.byref(wp))

    @staticmethod
    def _wparams(params):
        arr = (WorkloadParam * len(params))()
        for i, (name, v) in enumerate(params):
            arr[i].data_type = WP_UINT64
            arr[i].name = name.encode()
            arr[i].u_param = v
        wp = WorkloadParams(arr, len(params))
        return wp, arr

    def create(self, bench, params, sample_counts=(1, 1)):
        wp, _keep = self._wparams(params)
        hb = Handle()
        self.chk(self.L.createBenchmark(self.engine, bench["handle"], C.byref(wp), C.byref(hb)))
        concrete = BenchmarkDescriptor.from_buffer_copy(bench["desc"])
        if concrete.category == OFFLINE:
            for i, c in enumerate(sample_counts):
                concrete.cat_params.offline.data_count[i] = c
        self.chk(self.L.initBenchmark(hb, C.byref(concrete)))
        return hb

    def destroy(self, h):
        self.L.destroyHandle(h)

    def close(self):
        if self.engine.p:
            self.L.destroyHandle(self.engine)
            self.engine = Handle()

    # ---- data plumbing ----
    @staticmethod
    def pack(operands):
        """operands: list (one per op parameter) of 2-D numpy arrays [samples, n] -> DataPackCollection (+ keep-alives)."""
        keep = []
        packs = (DataPack * len(operands))()
        for i, arr in enumerate(operands):
            arr = np.ascontiguousarray(arr)
            bufs = (NativeDataBuffer * arr.shape[0])()
            for s in range(arr.shape[0]):
                row = arr[s]
                bufs[s].p = row.ctypes.data
                bufs[s].size = row.nbytes
                bufs[s].tag = 0
            packs[i].p_buffers = bufs
            packs[i].buffer_count = arr.shape[0]
            packs[i].param_position = i
            keep += [arr, bufs]
        return DataPackCollection(packs, len(operands)), keep + [packs]

    def run(self, hb, operands, out_n, out_dtype, indexers=None):
        """Full pipeline of one benchmark instance; returns results [b0*b1, out_n]."""
        L = self.L
        dpc, keep = self.pack(operands)
        h_plain, h_cipher, h_remote, h_out = Handle(), Handle(), Handle(), Handle()
        self.chk(L.encode(hb, C.byref(dpc), C.byref(h_plain)))
        self.chk(L.encrypt(hb, h_plain, C.byref(h_cipher)))
        self.chk(L.load(hb, C.byref(h_cipher), 1, C.byref(h_remote)))
        if indexers is None:
            indexers = [(0, o.shape[0]) for o in operands]
        pi = (ParameterIndexer * len(indexers))(*[ParameterIndexer(v, b) for v, b in indexers])
        self.chk(L.operate(hb, h_remote, pi, len(indexers), C.byref(h_out)))
        local = (Handle * 2)()
        self.chk(L.store(hb, h_out, local, 2))
        assert local[1].p is None and local[1].size == 0  # excess handles are zero-filled (ckks eltwise .cpp:297-298)
        h_dec = Handle()
        self.chk(L.decrypt(hb, local[0], C.byref(h_dec)))
        n_res = int(np.prod([b for _, b in indexers]))
        res = np.zeros((n_res, out_n), dtype=out_dtype)
        out_pack, keep2 = self.pack([res])
        self.chk(L.decode(hb, h_dec, C.byref(out_pack)))
        for h in (h_plain, h_cipher, h_remote, h_out, local[0], h_dec):
            L.destroyHandle(h)
        return res
