"""HEBench API-Bridge boundary without a GPU: library loads, exports every bridge symbol, registers the
descriptors with the reference's contents, validates arguments with the reference's error codes, does the
client-side steps on the host, and fails loudly at load() (host->HBM boundary) when no device exists."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from hebench_harness import (Backend, BridgeError, ECODE_CRITICAL, ECODE_INVALID_ARGS, LATENCY, OFFLINE, SCHEME_BFV, SCHEME_CKKS, W_ADD, W_DOT, W_MUL, W_LOGREG3, W_MATMUL,
                             DT_FLOAT64, DT_INT64, Handle)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def backend():
    be = importlib.import_module("reference-seal-backend_amd")
    if not os.path.exists(be.LIB_PATH):
        be.build()
    b = Backend(be.LIB_PATH)
    yield b
    b.close()


def test_exports_every_bridge_symbol():
    be = importlib.import_module("reference-seal-backend_amd")
    hdr = open(os.path.join(ROOT, "include", "hebench_api_bridge.h")).read()
    decl = re.findall(r"^(?:ErrorCode|uint64_t)\s+(\w+)\(", hdr, re.M)
    assert len(decl) == 20, decl
    L = C.CDLL(be.LIB_PATH)
    assert not [s for s in decl if not hasattr(L, s)]


def test_engine_registration(backend):
    assert backend.scheme_name(SCHEME_CKKS) == "CKKS" and backend.scheme_name(SCHEME_BFV) == "BFV"
    assert backend.security_name(SCHEME_CKKS, 0) == "128 bits"  # seal_engine.cpp:105
    bs = backend.benchmarks()
    got = sorted((b["desc"].workload, b["desc"].scheme, b["desc"].category) for b in bs)
    want = sorted([(w, s, c) for w in (W_ADD, W_MUL, W_DOT) for s in (SCHEME_BFV, SCHEME_CKKS) for c in (LATENCY, OFFLINE)] + [(W_MATMUL, SCHEME_BFV, LATENCY)] * 3 + [(W_MATMUL, SCHEME_CKKS, LATENCY)] * 3
                  + [(W_LOGREG3, SCHEME_CKKS, LATENCY), (W_LOGREG3, SCHEME_CKKS, OFFLINE)])
    assert got == want and len(bs) == 20  # all 20 of the reference's descriptors (seal_engine.cpp:108-151)
    # ... in the reference's order: a harness configuration file addresses a benchmark by its position in this list
    order = [(b["desc"].workload, b["desc"].scheme, b["desc"].category, b["desc"].other) for b in bs]
    B, K = SCHEME_BFV, SCHEME_CKKS
    assert order == ([(w, s, c, 0) for w in (W_ADD, W_MUL, W_DOT) for c in (LATENCY, OFFLINE) for s in (B, K)]
                     + [(W_MATMUL, B, LATENCY, 1), (W_MATMUL, K, LATENCY, 1), (W_MATMUL, B, LATENCY, 0), (W_MATMUL, K, LATENCY, 0), (W_MATMUL, B, LATENCY, 2), (W_MATMUL, K, LATENCY, 2)]
                     + [(W_LOGREG3, K, LATENCY, 1), (W_LOGREG3, K, OFFLINE, 1)])
    for c in (LATENCY, OFFLINE):
        lr = backend.find(W_LOGREG3, SCHEME_CKKS, c)
        assert lr["desc"].other == 1 and lr["desc"].data_type == DT_FLOAT64  # LogRegOtherID
        assert lr["defaults"][0] == [("n", 16), ("PolyModulusDegree", 16384), ("MultiplicativeDepth", 6), ("CoefficientModulusBits", 45), ("ScaleBits", 45),
                                     ("NumThreads", 0)]
        if c == OFFLINE:
            assert list(lr["desc"].cat_params.offline.data_count)[:3] == [1, 1, 0]
    assert sorted((b["desc"].scheme, b["desc"].other) for b in bs if b["desc"].workload == W_MATMUL) == sorted(
        (s, o) for s in (SCHEME_BFV, SCHEME_CKKS) for o in (0, 1, 2))  # MatMultVal, CipherBatchAxis, Row
    cba = sorted((b["desc"].scheme, b["defaults"][0][4:7]) for b in bs if b["desc"].workload == W_MATMUL and b["desc"].other == 1)
    assert cba == sorted([(SCHEME_BFV, [("MultiplicativeDepth", 3), ("CoefficientModulusBits", 40), ("PlainModulusBits", 20)]),
                          (SCHEME_CKKS, [("MultiplicativeDepth", 3), ("CoefficientModulusBits", 45), ("ScaleBits", 45)])])
    mv = sorted((b["desc"].scheme, b["defaults"][0][4:7]) for b in bs if b["desc"].workload == W_MATMUL and b["desc"].other == 0)  # MatMultValOtherID
    assert mv == sorted([(SCHEME_BFV, [("MultiplicativeDepth", 2), ("CoefficientModulusBits", 40), ("PlainModulusBits", 20)]),
                         (SCHEME_CKKS, [("MultiplicativeDepth", 2), ("CoefficientMudulusBits", 45), ("ScaleBits", 45)])])  # sic: ckks matmultval .cpp:50
    mm = [b for b in bs if b["desc"].workload == W_MATMUL and b["desc"].other == 2 and b["desc"].scheme == SCHEME_BFV][0]  # MatMultRowOtherID
    mc = [b for b in bs if b["desc"].workload == W_MATMUL and b["desc"].other == 2 and b["desc"].scheme == SCHEME_CKKS][0]
    assert mc["defaults"][0][4:7] == [("MultiplicativeDepth", 3), ("CoefficientModulusBits", 45), ("ScaleBits", 45)]
    assert mm["defaults"][0] == [("rows_M0", 10), ("cols_M0", 9), ("cols_M1", 8), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3),
                                 ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0)]
    # algorithm name / description rows, as the reference's headers define them (they name rows of the harness's report)
    algo = {}
    for b in bs:
        txt = backend.description_text(b, b["defaults"][0])
        row = [r for r in txt.splitlines() if r.startswith(", Algorithm, ")][0]
        algo[(b["desc"].workload, b["desc"].other)] = row
    assert algo[(W_MATMUL, 0)] == ", Algorithm, MatMultVal, One matrix row per ciphertext, Encode transposes second matrix"
    assert algo[(W_MATMUL, 1)] == ", Algorithm, CipherBatchAxis, One matrix element per ciphertext"
    assert algo[(W_MATMUL, 2)] == ", Algorithm, MatMulRow, "
    assert algo[(W_LOGREG3, 1)] == ", Algorithm, HornerPolyEval, Horner method for polynomial evaluation, single input vector per ciphertext"
    assert algo[(W_ADD, 0)] == ", Algorithm, Vector, One vector per ciphertext"
    for b in bs:
        d = b["desc"]
        assert d.cipher_param_mask == 0xFFFFFFFF and d.security == 0 and d.other in ((0, 1, 2) if d.workload == W_MATMUL else (1,) if d.workload == W_LOGREG3 else (0,))
        assert d.data_type == (DT_FLOAT64 if d.scheme == SCHEME_CKKS else DT_INT64)
        if d.category == LATENCY:
            assert d.cat_params.latency.warmup_iterations_count == 1 and d.cat_params.min_test_time_ms == 0
    # default workload parameters: SURVEY.md App. C
    ck = backend.find(W_MUL, SCHEME_CKKS, OFFLINE)["defaults"][0]
    # exactly the reference's six parameters in the reference's order (ckks eltwise .h:31-42): NumDevices is an optional extra, not a default
    assert ck == [("n", 1000), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 45), ("ScaleBits", 45), ("NumThreads", 0)]
    dot = backend.find(W_DOT, SCHEME_CKKS, LATENCY)["defaults"][0]
    assert dot[0] == ("n", 100) and dot[3] == ("CoefficientModulusBits", 40)
    bfv = backend.find(W_ADD, SCHEME_BFV, LATENCY)["defaults"][0]
    assert bfv[3:5] == [("CoefficientModulusBits", 40), ("PlainModulusBits", 20)]
    assert backend.find(W_DOT, SCHEME_BFV, OFFLINE)["defaults"][0][3] == ("CoefficientModulusBits", 45)


def test_description_text(backend):
    b = backend.find(W_ADD, SCHEME_CKKS, OFFLINE)
    txt = backend.description_text(b, [("n", 10), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3), ("CoefficientModulusBits", 45),
                                       ("ScaleBits", 45), ("NumThreads", 0)])
    assert ", , Poly modulus degree, 8192" in txt and ", , Coefficient Modulus, 60, 45, 45, 60" in txt and ", , Scale, 2^45" in txt
    assert ", Algorithm, Vector, One vector per ciphertext" in txt
    # the reference's rows in the reference's order (ckks eltwise .cpp:104-112); the device row comes after them
    rows = [r for r in txt.splitlines() if r.startswith(",")]
    assert [r for r in rows if r.startswith(", Number of threads, ")], txt
    assert txt.index(", Algorithm, ") < txt.index(", Number of threads, ") < txt.index(", Device, ")
    lat = backend.description_text(backend.find(W_ADD, SCHEME_CKKS, LATENCY), [("n", 10), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3),
                                                                             ("CoefficientModulusBits", 45), ("ScaleBits", 45), ("NumThreads", 7)])
    assert ", Number of threads, 1" in lat  # Latency forces one thread (.cpp:98-99)
    off = backend.description_text(b, [("n", 10), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3), ("CoefficientModulusBits", 45),
                                       ("ScaleBits", 45), ("NumThreads", 7)])
    assert ", Number of threads, 7" in off
    assert ", Number of devices" not in off  # a reference-style parameter set (six parameters) is described as the reference describes it
    dev = backend.description_text(b, [("n", 10), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3), ("CoefficientModulusBits", 45),
                                       ("ScaleBits", 45), ("NumThreads", 0), ("NumDevices", 1)])
    assert dev.rstrip().endswith(", Number of devices, 1")
    bfv = backend.description_text(backend.find(W_ADD, SCHEME_BFV, OFFLINE), [("n", 10), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3),
                                                                            ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0)])
    assert ", , Plain Text Modulus Bits, 20" in bfv  # bfv eltwise .cpp:111


def test_argument_validation_codes(backend):
    b = backend.find(W_ADD, SCHEME_CKKS, LATENCY)
    base = [("n", 10), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 45), ("ScaleBits", 45), ("NumThreads", 0)]
    with pytest.raises(BridgeError) as ei:  # n <= 0 (ckks eltwise .cpp:129-131)
        backend.create(b, [("n", 0)] + base[1:])
    assert ei.value.code == ECODE_INVALID_ARGS and "Vector size must be greater than 0." in str(ei.value)
    with pytest.raises(BridgeError) as ei:  # n > slot count (.cpp:152-155)
        backend.create(b, [("n", 5000)] + base[1:])
    assert ei.value.code == ECODE_INVALID_ARGS and "cannot be greater than 4096" in str(ei.value)
    with pytest.raises(BridgeError) as ei:  # invalid parameters -> code 2, as HEBSEAL_ECODE_SEAL_ERROR (seal_context.cpp:94-97)
        backend.create(b, [("n", 10), ("PolyModulusDegree", 4096), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 45), ("ScaleBits", 45), ("NumThreads", 0)])
    assert ei.value.code == 2
    assert backend.error_description(2) and backend.error_description(3)
    hb = backend.create(b, base)
    # pack_count != 2 (.cpp:165-167)
    dpc, keep = backend.pack([np.zeros((1, 10))])
    h = Handle()
    code = backend.L.encode(hb, C.byref(dpc), C.byref(h))
    assert code == ECODE_INVALID_ARGS and "Expected 2" in backend.last_error()
    backend.destroy(hb)


def test_client_side_then_loud_failure_at_load(backend):
    be = importlib.import_module("reference-seal-backend_amd")
    if be.device_count() > 0:
        pytest.skip("a GPU is present")
    b = backend.find(W_ADD, SCHEME_CKKS, OFFLINE)
    hb = backend.create(b, [("n", 10), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 2), ("CoefficientModulusBits", 45), ("ScaleBits", 45),
                            ("NumThreads", 0)], (2, 2))
    rng = np.random.default_rng(0)
    dpc, keep = backend.pack([rng.uniform(-1, 1, (2, 10)), rng.uniform(-1, 1, (2, 10))])
    hp, hc, hr = Handle(), Handle(), Handle()
    backend.chk(backend.L.encode(hb, C.byref(dpc), C.byref(hp)))   # host: CKKS encoder
    backend.chk(backend.L.encrypt(hb, hp, C.byref(hc)))            # host: encryptor
    code = backend.L.load(hb, C.byref(hc), 1, C.byref(hr))         # host -> HBM: must fail, loudly, with the device code
    assert code == 3 and "no CPU fallback" in backend.last_error()
    code = backend.L.load(hb, C.byref(hc), 2, C.byref(hr))         # count != 1 (.cpp:279-282)
    assert code == ECODE_INVALID_ARGS
    for h in (hp, hc):
        backend.destroy(h)
    backend.destroy(hb)


def test_matmult_row_encode_errors_are_the_references(backend):
    """Error text and code of the row-major MatMult's encode(), as the reference words them (bfv row .cpp:190-204: the text names
    parameter 0 for either operand; an undersized sample is a CRITICAL error, a missing one INVALID_ARGS)."""
    bench = [b for b in backend.benchmarks() if b["desc"].workload == W_MATMUL and b["desc"].other == 2 and b["desc"].scheme == SCHEME_BFV][0]
    hb = backend.create(bench, [("rows_M0", 4), ("cols_M0", 3), ("cols_M1", 2), ("PolyModulusDegree", 8192), ("MultiplicativeDepth", 3),
                                ("CoefficientModulusBits", 40), ("PlainModulusBits", 20), ("NumThreads", 0)])
    a, b = np.zeros((1, 12), dtype=np.int64), np.zeros((1, 6), dtype=np.int64)
    h = Handle()
    dpc, keep = backend.pack([a, np.zeros((1, 5), dtype=np.int64)])  # second operand one value short
    with pytest.raises(BridgeError) as ei:
        backend.chk(backend.L.encode(hb, C.byref(dpc), C.byref(h)))
    assert ei.value.code == ECODE_CRITICAL and "Insufficient data for parameter 0 sample." in str(ei.value)
    dpc, keep = backend.pack([a, np.zeros((0, 6), dtype=np.int64)])  # no sample for the second operand
    with pytest.raises(BridgeError) as ei:
        backend.chk(backend.L.encode(hb, C.byref(dpc), C.byref(h)))
    assert ei.value.code == ECODE_INVALID_ARGS
    assert "Latency test requires, at least, 1 sample per operation parameter. None found for operation parameter 0." in str(ei.value)
    dpc, keep = backend.pack([a, b])  # and the well-formed call goes through (host-side encoders without a GPU)
    backend.chk(backend.L.encode(hb, C.byref(dpc), C.byref(h)))
    backend.L.destroyHandle(h)
    backend.destroy(hb)


def test_harness_constants_match_the_library(backend):
    """The enumerators, sizes and offsets this harness hard-codes are the ones the library was compiled with
    (he355_bridge_abi, csrc/bridge/abi_check.cpp): a header change -- including a build against the real api-bridge
    header with other numbers -- shows up here instead of silently shifting workload ids."""
    import json
    import hebench_harness as hh
    be = importlib.import_module("reference-seal-backend_amd")
    L = C.CDLL(be.LIB_PATH)
    L.he355_bridge_abi.restype = C.c_uint64
    L.he355_bridge_abi.argtypes = [C.c_char_p, C.c_uint64]
    need = L.he355_bridge_abi(None, 0)
    buf = C.create_string_buffer(need)
    assert L.he355_bridge_abi(buf, need) == need
    abi = json.loads(buf.value.decode())
    assert abi["api_version"] == [0, 8, 0]  # cmake/third-party/API_BRIDGE.version:1-5
    w, dt, cat = abi["workload"], abi["data_type"], abi["category"]
    assert (hh.W_MATMUL, hh.W_MUL, hh.W_ADD, hh.W_DOT, hh.W_LOGREG3) == (w["MatrixMultiply"], w["EltwiseMultiply"], w["EltwiseAdd"], w["DotProduct"],
                                                                       w["LogisticRegression_PolyD3"])
    assert (hh.DT_INT64, hh.DT_FLOAT64) == (dt["Int64"], dt["Float64"]) and (hh.LATENCY, hh.OFFLINE) == (cat["Latency"], cat["Offline"])
    assert hh.WP_UINT64 == abi["workload_param_type"]["UInt64"]
    assert (hh.SCHEME_CKKS, hh.SCHEME_BFV) == (abi["scheme"]["CKKS"], abi["scheme"]["BFV"])
    assert (hh.MAX_BUF, hh.MAX_OP) == (abi["max_buffer_size"], abi["max_op_params"]) and abi["max_category_params"] == 2 * hh.MAX_OP
    assert (hh.ECODE_INVALID_ARGS, hh.ECODE_CRITICAL) == (abi["ecode_invalid_args"], abi["ecode_critical_error"])
    so = abi["sizeof"]
    for name, cls in (("Handle", hh.Handle), ("DataPack", hh.DataPack), ("DataPackCollection", hh.DataPackCollection), ("ParameterIndexer", hh.ParameterIndexer),
                      ("CategoryParams", hh.CategoryParams), ("BenchmarkDescriptor", hh.BenchmarkDescriptor), ("WorkloadParam", hh.WorkloadParam),
                      ("WorkloadParams", hh.WorkloadParams)):
        assert C.sizeof(cls) == so[name], name
    oo = abi["offsetof"]
    for f in ("cat_params", "cipher_param_mask", "scheme", "security", "other"):
        assert getattr(hh.BenchmarkDescriptor, f).offset == oo["BenchmarkDescriptor." + f], f
    assert hh.WorkloadParam.name.offset == oo["WorkloadParam.name"] and hh.WorkloadParam.v.offset == oo["WorkloadParam.u_param"]
