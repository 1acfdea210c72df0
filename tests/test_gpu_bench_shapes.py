"""The shapes `bench.py` runs, under the driver's own `pytest -m gpu`, bit for bit against the oracle.

Kernel selection in the library goes by batch and ring size (thresholds at 17 / 25 ciphertexts, 512 / 1536 / 4096 blocks, whole groups per
launch: DESIGN.md 5.2-5.3), so the bench shapes are code paths of their own: configs[3] at n = 64 runs `k_k3_dual8` and the small-grid
rules that n = 8 does not, configs[4]'s 64-ciphertext 127-step rotate_sum at N = 2^15 forms its level sums inside the fused `k_k3`.  Every
`bench.WORKLOADS` entry is instantiated here at its default batch exactly as `bench.py` does (same class, same `setup()` / `step()`), and
the FIRST and the LAST result rows of the step are compared with the workload's own checker leg (the oracle).

Reference loops: /root/reference/src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:325-336 (batch loop),
.../ckks/seal_ckks_dot_product_benchmark.cpp:325-330, .../bfv/seal_bfv_matmult_row_benchmark.cpp:512-533,
.../ckks/seal_ckks_matmultval_benchmark.cpp:253-255 (multiply -> relinearize -> rescale).
"""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

# The path-counter assertions describe the LIBRARY'S OWN choice of shape: under a run-time setting that pins another one (tools/test_matrix.sh
# runs this module under all of them) only the bits are held to the oracle.
_SHAPE_ENV = ("HE355_K3_FUSE", "HE355_CHUNK", "HE355_LATENCY_MAX", "HE355_LEVEL_WALK", "HE355_LDS_MAX", "HE355_DUAL_ENGINE", "HE355_FORCE_U64")
DEFAULT_SHAPES = not any(os.environ.get(k) for k in _SHAPE_ENV)


@pytest.fixture(scope="module")
def be():
    mod = importlib.import_module("reference-seal-backend_amd")
    if mod.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests need an MI355X (the backend has no CPU fallback)")
    return mod


def _instantiate(be, name, batch=None, b1=0):
    import bench
    W = bench.WORKLOADS[name]
    if b1:
        W = type(W.__name__ + f"_b1_{b1}", (W,), {"b1": b1})
    bits = W.bits or be.chain_bits(W.depth, W.coeff_bits)
    ctx = be.Context(be.SCHEME_CKKS if W.scheme == "ckks" else be.SCHEME_BFV, W.N, bit_sizes=bits, plain_bits=W.plain_bits, device=0)
    shard = bench.load_sharding().shard_outer_product(batch or W.default_batch, W.b1, 1, 0)
    wl = W(be, ctx, shard, None)
    wl.setup()
    return W, bits, ctx, wl


def _oracle_ctx(ho, W, bits, ctx):
    o = ho.Context(ho.SCHEME_CKKS if W.scheme == "ckks" else ho.SCHEME_BFV, W.N, bit_sizes=bits, plain_bits=W.plain_bits)
    assert [int(q) for q in o.moduli] == [int(q) for q in ctx.moduli]
    return o


def _check_first_and_last(ho, o, wl, rows):
    """`rows` operand-0 rows at the head and at the tail of the batch: the step's results there == the oracle's"""
    threads = max(1, min(8, ho.lib().ho_max_threads()))
    for first in sorted({0, wl.rows - rows}):
        wl.first_row = first
        _, want = wl.checker(ho, o, rows, threads, 1)
        got = wl.result_rows(rows * wl.b1)
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"{wl.name}: results of operand-0 rows {first}..{first + rows - 1} differ from the oracle"
    wl.first_row = 0


# (workload, batch, b1, rows held to the oracle at each end)
SHAPES = [
    ("mul_relin_rescale", 1024, 0, 2),   # BASELINE configs[2]: the judged line, one chunk of 1024
    ("mul_relin", 1024, 0, 1),           # `metric` as worded (no rescale)
    ("eltwise_mul", 256, 0, 2),          # configs[1] as 256 x 1
    ("eltwise_mul", 16, 16, 1),          # configs[1] as 16 x 16 (operand 1 batched)
    ("dot", 64, 0, 1),                   # configs[3] at the bench's n = 64: k_k3_dual8, fuse_pays, the 4096-block rule
    ("bfv_add", 4096, 0, 2),             # configs[0] at the reference defaults
]


@pytest.mark.parametrize("name,batch,b1,rows", SHAPES, ids=[f"{n}-{b}x{max(1, x)}" for n, b, x, _ in SHAPES])
def test_bench_shape_first_and_last_rows_equal_the_oracle(be, oracle, name, batch, b1, rows):
    import bench
    assert bench.WORKLOADS[name].default_batch == batch or b1, "the test must follow bench.py's default batch"
    W, bits, ctx, wl = _instantiate(be, name, batch, b1)
    try:
        assert wl.n == batch * max(1, b1)
        ctx.path_stats(reset=True)
        wl.step()
        ctx.sync()
        st = ctx.path_stats()
        if DEFAULT_SHAPES and name in ("mul_relin_rescale", "mul_relin"):
            assert st["ks_fused"] >= 1 and st["ks_unfused"] == 0 and st["ks_latency"] == 0, st  # the headline is the fused throughput shape
        if DEFAULT_SHAPES and name == "dot":
            assert st["ks_fused"] == 13 and st["ks_latency"] == 0 and st["ks_unfused"] == 0, st  # 1 relinearization + 12 rotations, all fused at n = 64
        _check_first_and_last(oracle, _oracle_ctx(oracle, W, bits, ctx), wl, rows)
    finally:
        ctx.close()


def test_bfv_matmul_bench_shape_level_sums_inside_the_key_switch(be, oracle):
    """configs[4] as bench.py runs it: 64 row-pair ciphertexts, 127 rotations as a 127-node trie walked level by level.  The level sums must be
    formed inside the fused k_k3 (K3Args::og_stride / KsGroups::sum_out -- the round-5 path); first and last result ciphertext == the oracle's
    unshared loop; and the same bits with the chunk (a) a few whole groups per launch -- several level-sum launches per level, g_op_offset != 0
    -- and (b) below the group size, where k_sum_groups adds the level."""
    W, bits, ctx, wl = _instantiate(be, "bfv_matmul")
    try:
        assert wl.n == 64
        ctx.path_stats(reset=True)
        wl.step()
        ctx.sync()
        st = ctx.path_stats()
        assert wl.key_switches == 127
        full = wl.result.download((64, 2, wl.L, wl.N)).copy()
        _check_first_and_last(oracle, _oracle_ctx(oracle, W, bits, ctx), wl, 1)
        if not DEFAULT_SHAPES:
            return  # (another shape was pinned from outside: the bits above are what this run can hold)
        levels = st["level_sums_in_k3"] + st["level_sums_by_kernel"]
        assert levels >= 4 and st["level_sums_by_kernel"] == 0 and st["level_sums_in_k3"] == levels, st  # (the NAF trie of j * 128, j < 128, is 4 deep)
        launches_default = st["level_sum_launches_in_k3"]
        assert launches_default >= levels, st  # (a level of 13-50 nodes x 64 ciphertexts is cut at the default chunk of 1024 too: whole groups per launch)
        # (a) chunk = 5 groups of 64: levels of 13-50 nodes are cut into several launches, each with its own g_op_offset, all summing in k_k3
        ctx.set_chunk(5 * 64)
        ctx.path_stats(reset=True)
        wl.step()
        ctx.sync()
        st = ctx.path_stats()
        assert st["level_sums_by_kernel"] == 0 and st["level_sums_in_k3"] == levels and st["level_sum_launches_in_k3"] > launches_default, st
        assert np.array_equal(wl.result.download((64, 2, wl.L, wl.N)), full), "level sums in k_k3 with several launches per level changed the result"
        # (b) chunk below the group size: no launch holds a whole group, k_sum_groups forms every level's sum
        ctx.set_chunk(32)
        ctx.path_stats(reset=True)
        wl.step()
        ctx.sync()
        st = ctx.path_stats()
        assert st["level_sums_in_k3"] == 0 and st["level_sums_by_kernel"] == levels, st
        assert np.array_equal(wl.result.download((64, 2, wl.L, wl.N)), full), "k_sum_groups and the in-kernel level sum disagree"
    finally:
        ctx.close()
