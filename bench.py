#!/usr/bin/env python3
"""Benchmarks of the hot path on MI355X, one JSON line per run (rank 0).

Default (the judged line) = BASELINE.json configs[2], the configuration `metric` is quoted on:
CKKS multiply -> relinearize -> rescale, N=2^15, L=16, batch 1024 per GPU.  One op = one result ciphertext of the per-pair
pipeline (the sequence at /root/reference/src/benchmarks/ckks/seal_ckks_matmultval_benchmark.cpp:253-255 applied per batch
element with the HEBench outer-product indexing of seal_ckks_element_wise_benchmark.cpp:322-336).  A "step" is one pass of
that pipeline over the rank's resident shard of the batch.

  --config mul_relin_rescale | eltwise_mul | dot | bfv_matmul | bfv_add     (BASELINE.json configs[2], [1], [3], [4], [0])
  --scaling weak | strong      weak: --batch results PER GPU, global batch = batch x N;  strong: --batch is the GLOBAL batch, cut into
                               contiguous blocks of operand 0 (reference-seal-backend_amd/sharding.py).  Not given (how the driver runs it):
                               N = 1 is one run (the two coincide); N > 1 times BOTH in the one invocation -- `value` / `scaling` = strong
                               at the global batch `metric` names (north_star: "batch = 1024 ... >= 6x at 8 GPUs"), and a `weak` block
                               (value, ms_per_step, per_rank_ms) for --batch results per GPU beside it
  --force-dist                 (or HE355_BENCH_FORCE_DIST=1) at --gpus 1: start the one rank under torch.distributed.run and initialise the
                               collective backend (nccl = RCCL) anyway -- the multi-GPU branch as far as a 1-GPU box can take it
One process per GPU (torch.distributed.run), no data-path collective: results are independent.  Every rank
builds the same evaluation keys on its own device from the shared seed (the generators are counter-based: a pure function of
seed and index) and fills its shard of the global operand array with the values the whole array holds there
(he355_fill_uniform_at), so the job computes the same global batch at every world size.

Timing: W untimed warm-up steps, barrier + synchronize, K timed steps, barrier + synchronize, MAX over ranks; `value` =
results of all ranks / that time.  Inputs and keys are resident in HBM before the timed region.  Rank 0 at N=1 also times the
oracle (CPU port, `cpu_baseline`) on a bounded sample; every rank pushes a small sample of ITS shard through the checker and
compares bit for bit (`parity`): a mismatch on any rank makes the run exit non-zero.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md
SEED_A, SEED_B, SEED_RELIN, SEED_GALOIS = 1234, 99, 7, 100


def load_sharding():
    import importlib.util
    spec = importlib.util.spec_from_file_location("he355_sharding", os.path.join(ROOT, "reference-seal-backend_amd", "sharding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules.setdefault("he355_sharding", mod)
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------------------------------------------------------
# Workloads.  Each owns: the context parameters, the resident operands of this rank's shard, step(), the algorithmic
# bytes per result, and the checker leg (oracle on a bounded sample of the shard).
# ---------------------------------------------------------------------------------------------------------------
class Workload:
    name = ""
    scheme = "ckks"
    N = 32768
    bits: list = []
    plain_bits = 0
    b1 = 1                 # operand-1 batch (replicated on every rank)
    default_batch = 1024   # operand-0 batch per GPU (weak) / global (strong)
    default_cpu_sample = 0
    parity_sample = 2

    def __init__(self, be, ctx, shard, args):
        self.be, self.ctx, self.shard, self.args = be, ctx, shard, args
        self.L, self.K = ctx.L, ctx.K
        self.rows = shard.a_count            # operand-0 rows of this rank
        self.n = shard.n_results             # results of this rank per step
        self.pm = list(range(self.L))

    # operand 0: this rank's rows of the global array; operand 1: replicated
    def fill_operands(self, size=2):
        c, L, N = self.ctx, self.L, self.N
        self.d_a = c.alloc(max(1, self.rows) * size * L * N)
        self.d_b = c.alloc(self.b1 * size * L * N)
        c.fill_uniform(self.d_a, self.rows * size * L, self.pm, SEED_A, first_poly=self.shard.a_base * size * L)
        c.fill_uniform(self.d_b, self.b1 * size * L, self.pm, SEED_B)
        self.ix = self.be.Context.outer(0, self.rows, 0, self.b1)

    first_row = 0  # checker legs: the operand-0 row the sample starts at (0: the head of the shard; tests also hold the LAST rows to the oracle)

    def host_operands(self, rows, size=2):
        """`rows` rows of this rank's operand 0 starting at self.first_row, and all of operand 1, as numpy arrays (checker input)"""
        L, N = self.L, self.N
        a = self.d_a.download_range(self.first_row * size * L * N, (rows, size, L, N))
        b = self.d_b.download_head((self.b1, size, L, N))
        return a, b

    def result_slab(self):
        """(device slab of the step's results, u64 words per result)"""
        raise NotImplementedError

    def result_rows(self, k):
        """k results starting at result self.first_row * b1"""
        buf, per = self.result_slab()
        return buf.download_range(self.first_row * self.b1 * per[0] * per[1] * self.N, (k,) + per + (self.N,))

    def key_bytes(self):
        return self.L * 2 * self.K * self.N * 8

    in_size, out_size, out_drop, n_keys = 2, 2, 0, 0  # ciphertext sizes in / out, residues dropped by the pipeline, evaluation keys touched

    def compulsory_bytes_per_op(self, b0, b1):
        """bytes that must cross HBM per result when every operand row and key is read once per batch and every result written once"""
        poly = self.N * 8
        ct_in, ct_out = self.in_size * self.L * poly, self.out_size * (self.L - self.out_drop) * poly
        return ct_out + ct_in / b1 + ct_in / b0 + self.n_keys * self.key_bytes() / (b0 * b1)

    def describe(self):
        raise NotImplementedError


class MulRelinRescale(Workload):
    """configs[2]: multiply -> relinearize_inplace -> rescale_to_next_inplace (ckks matmultval .cpp:253-255)"""
    name = "mul_relin_rescale"
    out_drop, n_keys = 1, 1
    N, depth, coeff_bits = 32768, 16, 45
    default_batch, default_cpu_sample = 1024, 256

    def setup(self):
        c, L, N = self.ctx, self.L, self.N
        self.fill_operands()
        self.d_out = c.alloc(max(1, self.n) * 2 * (L - 1) * N)
        c.set_relin_key_synthetic(SEED_RELIN)

    def step(self):
        self.ctx.multiply_relin(self.L, self.n, self.d_a, self.d_b, self.ix, self.d_out, rescale=True)

    def bytes_per_op(self, global_batch):
        """read 2 cts + write 1 ct at L-1 + the relin key once per batch = 24,780,800 B at batch 1024"""
        L, N = self.L, self.N
        return 2 * (2 * L * N * 8) + 2 * (L - 1) * N * 8 + self.key_bytes() / global_batch

    def result_slab(self):
        return self.d_out, (2, self.L - 1)

    def checker(self, ho, o, rows, threads, passes):
        """oracle on the first `rows` rows of the shard; returns (seconds per pass list, expected results)"""
        a, b = self.host_operands(rows)
        rk = synthetic_key_host(self.ctx, o, SEED_RELIN)
        o.batch_outer(ho.OP_MUL_RELIN_RESCALE, a[:min(rows, threads)], b, rk, threads=threads)  # warm-up (tables, pages)
        times, want = [], None
        for _ in range(passes):
            t0 = time.perf_counter()
            want = o.batch_outer(ho.OP_MUL_RELIN_RESCALE, a, b, rk, threads=threads)
            times.append(time.perf_counter() - t0)
        return times, want

    def describe(self):
        return (f"CKKS EltwiseMult + relinearize + rescale, N=2^15, depth 16 (L=16 data primes + 1 special) (BASELINE.json configs[2])")


class MulRelin(MulRelinRescale):
    """`metric` as literally worded: multiply -> relinearize_inplace, no rescale (size-2 result at level L)"""
    name = "mul_relin"
    out_drop = 0

    def setup(self):
        c, L, N = self.ctx, self.L, self.N
        self.fill_operands()
        self.d_out = c.alloc(max(1, self.n) * 2 * L * N)
        c.set_relin_key_synthetic(SEED_RELIN)

    def step(self):
        self.ctx.multiply_relin(self.L, self.n, self.d_a, self.d_b, self.ix, self.d_out, rescale=False)

    def bytes_per_op(self, global_batch):
        L, N = self.L, self.N
        return 3 * (2 * L * N * 8) + self.key_bytes() / global_batch  # read 2 cts, write 1 at the same level, the key once per batch

    def result_slab(self):
        return self.d_out, (2, self.L)

    def checker(self, ho, o, rows, threads, passes):
        a, b = self.host_operands(rows)
        rk = synthetic_key_host(self.ctx, o, SEED_RELIN)
        o.batch_outer(ho.OP_MUL_RELIN, a[:min(rows, threads)], b, rk, threads=threads)
        times, want = [], None
        for _ in range(passes):
            t0 = time.perf_counter()
            want = o.batch_outer(ho.OP_MUL_RELIN, a, b, rk, threads=threads)
            times.append(time.perf_counter() - t0)
        return times, want

    def describe(self):
        return "CKKS EltwiseMult + relinearize (no rescale), N=2^15, depth 16 (L=16 data primes + 1 special): BASELINE.json `metric` as worded"


class EltwiseMul(Workload):
    """configs[1]: Evaluator::multiply only (ckks eltwise .cpp:343), size-3 results"""
    name = "eltwise_mul"
    out_size = 3
    N, depth, coeff_bits = 16384, 8, 45
    default_batch, default_cpu_sample = 256, 256

    def setup(self):
        c, L, N = self.ctx, self.L, self.N
        self.fill_operands()
        self.d_out = c.alloc(max(1, self.n) * 3 * L * N)

    def step(self):
        self.ctx.multiply(self.L, self.n, self.d_a, self.d_b, self.ix, self.d_out)

    def bytes_per_op(self, global_batch):
        return 7 * self.L * self.N * 8  # read 4 polys, write 3 = 7,340,032 B

    def result_slab(self):
        return self.d_out, (3, self.L)

    def checker(self, ho, o, rows, threads, passes):
        a, b = self.host_operands(rows)
        times, want = [], None
        o.batch_outer(ho.OP_MUL, a[:min(rows, threads)], b, threads=threads)
        for _ in range(passes):
            t0 = time.perf_counter()
            want = o.batch_outer(ho.OP_MUL, a, b, threads=threads)
            times.append(time.perf_counter() - t0)
        return times, want

    def describe(self):
        return "CKKS EltwiseMult (multiply only, size-3 results), N=2^14, depth 8 (L=8) (BASELINE.json configs[1])"


class DotProduct(Workload):
    """configs[3]: multiply -> relinearize_inplace -> accumulateCKKS(4096) (ckks dot .cpp:325-330): 13 key switches per result"""
    name = "dot"
    n_keys = 13
    N, depth, coeff_bits = 32768, 16, 45
    count = 4096
    default_batch, default_cpu_sample = 64, 8
    parity_sample = 1

    def setup(self):
        c, L, N = self.ctx, self.L, self.N
        self.fill_operands()
        self.d_out = c.alloc(max(1, self.n) * 2 * L * N)
        self.d_tmp = c.alloc(max(1, self.n) * 2 * L * N)
        c.set_relin_key_synthetic(SEED_RELIN)
        self.steps_log = 12
        for k in range(self.steps_log):
            c.set_galois_key_synthetic(c.galois_elt(1 << k), SEED_GALOIS + k)

    def step(self):
        self.ctx.multiply_relin(self.L, self.n, self.d_a, self.d_b, self.ix, self.d_out)
        self.ctx.accumulate(self.L, self.n, self.d_out, self.count, self.d_tmp)

    def bytes_per_op(self, global_batch):
        """16 MiB in + 8 MiB out = 25,165,824 B + (relin + 12 Galois keys) once per batch"""
        L, N = self.L, self.N
        return 2 * (2 * L * N * 8) + 2 * L * N * 8 + 13 * self.key_bytes() / global_batch

    def result_slab(self):
        return self.d_out, (2, self.L)

    def checker(self, ho, o, rows, threads, passes):
        a, b = self.host_operands(rows)
        rk = synthetic_key_host(self.ctx, o, SEED_RELIN)
        gk = {o.galois_elt(1 << k): synthetic_key_host(self.ctx, o, SEED_GALOIS + k) for k in range(self.steps_log)}
        times, want = [], None
        for _ in range(passes):
            t0 = time.perf_counter()
            want = o.batch_outer(ho.OP_DOT, a, b, rk, gk, self.count, threads=threads)
            times.append(time.perf_counter() - t0)
        return times, want

    def describe(self):
        return ("CKKS DotProduct, vector length 4096: multiply + relinearize + 12 x (rotate 2^i + add), N=2^15, depth 16 (L=16), 13 key switches per "
                "result (BASELINE.json configs[3])")


class BfvMatMul(Workload):
    """configs[4]: BFV MatMultRow 128x128x128 (bfv row .cpp:486-539): per row-pair ciphertext multiply + relinearize + 127 rotate_rows
    (j * 128, NAF-decomposed: 355 key switches) + add.  One op = one result (row-pair) ciphertext; 64 per matrix product."""
    name = "bfv_matmul"
    n_keys = 29
    scheme = "bfv"
    N, bits, plain_bits = 32768, [60, 40, 40, 60], 20
    dim = 128
    default_batch, default_cpu_sample = 64, 2
    parity_sample = 1

    def setup(self):
        c, L, N = self.ctx, self.L, self.N
        self.fill_operands()
        n = max(1, self.n)
        self.c3, self.base, self.acc = c.alloc(n * 3 * L * N), c.alloc(n * 2 * L * N), c.alloc(n * 2 * L * N)
        c.set_relin_key_synthetic(SEED_RELIN)
        self.gk_steps = []
        k = 0
        while (1 << k) < N // 2:  # the default Galois key set: +-2^k row rotations (create_galois_keys, seal_context.cpp:69)
            self.gk_steps += [1 << k, -(1 << k)]
            k += 1
        for i, s in enumerate(self.gk_steps):
            c.set_galois_key_synthetic(c.galois_elt(s), SEED_GALOIS + i)
        self.spacers = (N // 2) // self.dim
        self.result = None

    def step(self):
        c, L, n = self.ctx, self.L, self.n
        c.bfv_multiply(L, n, self.d_a, self.d_b, self.ix, self.c3)
        c.relinearize(L, n, self.c3, self.base)
        # result = base; result += rotate_rows(base, j * spacers), j = 1 .. dim-1 (bfv row .cpp:519-531): all rotations start from `base`,
        # he355_rotate_sum key-switches every distinct NAF prefix once (bridge/matmult_row.cpp does the same)
        self.key_switches = c.rotate_sum(L, n, self.base, [j * self.spacers for j in range(1, self.dim)], self.acc)
        cur = self.acc
        self.result = cur

    def bytes_per_op(self, global_batch):
        """per result ciphertext: read A[i] and write the result (2 x 2 L N 8), + B and the keys (relin + 28 Galois) once per batch"""
        L, N = self.L, self.N
        ct = 2 * L * N * 8
        return 2 * ct + (ct + (1 + len(self.gk_steps)) * self.key_bytes()) / global_batch

    def result_slab(self):
        return self.result, (2, self.L)

    def checker(self, ho, o, rows, threads, passes):
        a, b = self.host_operands(rows)
        rk = synthetic_key_host(self.ctx, o, SEED_RELIN)
        gk = {o.galois_elt(s): synthetic_key_host(self.ctx, o, SEED_GALOIS + i) for i, s in enumerate(self.gk_steps)}
        times, want = [], None
        for _ in range(passes):
            t0 = time.perf_counter()
            want = o.bfv_matmul_rows(a, b[0], rk, gk, self.dim, threads=threads)
            times.append(time.perf_counter() - t0)
        return times, want

    def describe(self):
        ks = getattr(self, "key_switches", None)
        return ("BFV MatMul 128x128x128 (MatMultRow): 64 row-pair ciphertexts x (BEHZ multiply + relinearize + 127 rotate_rows + add), N=2^15, "
                "{60,40,40,60}; the 127 rotations are 313 NAF terms in the reference's loop, " + (f"{ks} Galois key switches here" if ks else "fewer here")
                + " (rotations with a common NAF prefix share it: he355_rotate_sum), + 1 relinearization per result ciphertext (BASELINE.json configs[4])")


class BfvAdd(Workload):
    """configs[0] at the reference's default parameters (N=8192, {60,40,60}: SURVEY.md 8d; the literal N=4096 single-modulus case is a
    parity test, tests/test_gpu_parity_bfv.py): Evaluator::add (bfv eltwise .cpp:322)"""
    name = "bfv_add"
    scheme = "bfv"
    N, bits, plain_bits = 8192, [60, 40, 60], 20
    default_batch, default_cpu_sample = 4096, 1024

    def setup(self):
        c, L, N = self.ctx, self.L, self.N
        self.fill_operands()
        self.d_out = c.alloc(max(1, self.n) * 2 * L * N)

    def step(self):
        self.ctx.add(self.L, 2, self.n, self.d_a, self.d_b, self.ix, self.d_out)

    def bytes_per_op(self, global_batch):
        return 3 * 2 * self.L * self.N * 8  # read 2 cts (operand 1 from cache after the first result), write 1

    def result_slab(self):
        return self.d_out, (2, self.L)

    def checker(self, ho, o, rows, threads, passes):
        a, b = self.host_operands(rows)
        times, want = [], None
        for _ in range(passes):
            t0 = time.perf_counter()
            want = o.batch_outer(ho.OP_ADD, a, b, threads=threads)
            times.append(time.perf_counter() - t0)
        return times, want

    def describe(self):
        return "BFV EltwiseAdd at the reference's defaults N=8192, {60,40,60} (L=2) (BASELINE.json configs[0])"


WORKLOADS = {w.name: w for w in (MulRelinRescale, MulRelin, EltwiseMul, DotProduct, BfvMatMul, BfvAdd)}


N_SIMD = 1024  # 256 CUs x 4 SIMDs


def committed_counters(config, b1, n_results):
    """(json, path) of the newest profiles/r*_<config>[_b1x<b1>]_kernel_bounds.json whose step had the same shape, else None"""
    import glob
    tag = config + (f"_b1x{b1}" if b1 > 1 else "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_kernel_bounds.json")), reverse=True):
        try:
            j = json.load(open(path))
        except (OSError, ValueError):
            continue
        if j.get("hbm_bytes_per_op"):
            return j, "profiles/" + os.path.basename(path)
    return None


def synthetic_key_host(ctx, o, seed):
    """The synthetic evaluation key he355_set_*_key_synthetic(seed) builds on the device, regenerated on the host for the checker:
    the same counter-based stream (he355_fill_uniform over the key's [L][2][K][N] polynomials, prime = polynomial index mod K)."""
    L, K, N = ctx.L, ctx.K, ctx.N
    buf = ctx.alloc(L * 2 * K * N)
    ctx.fill_uniform(buf, L * 2 * K, list(range(K)), seed)
    return buf.download((L, 2, K, N)).copy()


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: this process -- which has made no GPU call -- starts the N ranks
    as a child (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, one rank per GPU),
    relays their output (rank 0 prints the JSON line) and returns the child's exit code.  Under torch.distributed.run
    (WORLD_SIZE set) main() runs as a rank and this is never reached."""
    import subprocess
    # --standalone: torchrun owns the rendezvous (c10d store on a port it picks and holds itself), so no port can be taken between a
    # probe and the launch; --local-addr keeps it on the loopback interface (the container's hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_gpus) // n_gpus)))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def scalings_of(args, world):
    """The scalings one invocation times, primary first.  N = 1: one run (both scalings are the same job).  N > 1 without --scaling: BOTH --
    `value` answers north_star's strong question (the GLOBAL batch is --batch, 1024 for the headline, at every N: ">= 6x at 8 GPUs"), and a
    `weak` block beside it times --batch results PER GPU (the 1-GPU load on every GPU).  --scaling weak|strong times that one only."""
    if args.scaling:
        return [args.scaling]
    return ["weak"] if world == 1 else ["strong", "weak"]


def dry_run(args, rank, world, dist, torch) -> int:
    """The N>1 path without a GPU: rendezvous over gloo, the shard every rank would own under every scaling the real run would time, one
    JSON line from rank 0 (the primary scaling's shards at the top level, every scaling under `runs`)."""
    use_dist = world > 1 or args.force_dist
    if use_dist:
        dist.init_process_group(backend="gloo")
    W = WORKLOADS[args.config]
    if args.b1 > 0:
        W = type(W.__name__ + f"_b1_{args.b1}", (W,), {"b1": args.b1})
    batch = args.batch or W.default_batch
    runs = []
    for scaling in scalings_of(args, world):
        global_b0 = batch * world if scaling == "weak" else batch
        sh = load_sharding().shard_outer_product(global_b0, W.b1, world, rank)
        mine = torch.tensor([rank, sh.a_base, sh.a_count, sh.n_results], dtype=torch.int64)
        rows = [mine]
        if use_dist:
            rows = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(rows, mine)
            dist.barrier()
        runs.append({"scaling": scaling, "global_b0": global_b0,
                     "shards": [{"rank": int(r[0]), "a_base": int(r[1]), "a_count": int(r[2]), "n_results": int(r[3])} for r in rows]})
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "config": args.config, "scaling": runs[0]["scaling"], "global_b0": runs[0]["global_b0"],
                          "b1": W.b1, "shards": runs[0]["shards"], "runs": runs}))
    if use_dist:
        dist.destroy_process_group()
    return 0



def sample_power(props, run_steps, step_seconds, window_seconds=1.5):
    """Median / maximum package power (PPT) and core clock the device reports while `run_steps(k)` runs k steps, about `window_seconds` of them, read from
    the hwmon directory of the device's PCI function; None when the files are not there or not readable (the line then carries no `power`)."""
    import glob
    import threading
    try:
        bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        dirs = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        if not dirs or not os.path.exists(os.path.join(dirs[0], "power1_input")):
            return None
        hw = dirs[0]

        def read(name):
            with open(os.path.join(hw, name)) as f:
                return int(f.read().strip())

        cap = read("power1_cap") / 1e6 if os.path.exists(os.path.join(hw, "power1_cap")) else None
        idle = read("power1_input") / 1e6
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                try:
                    samples.append((read("power1_input") / 1e6, read("freq1_input") / 1e6 if os.path.exists(os.path.join(hw, "freq1_input")) else None))
                except OSError:
                    pass
                stop.wait(0.02)

        steps = max(3, min(20000, int(window_seconds / max(step_seconds, 1e-6))))
        th = threading.Thread(target=poll, daemon=True)
        t0 = time.perf_counter()
        th.start()
        run_steps(steps)  # queued back to back like the timed steps, one synchronisation at the end
        window = time.perf_counter() - t0
        stop.set()
        th.join(timeout=2.0)
        busy = samples[len(samples) // 3:]  # the telemetry is a moving average: the first third of the window is its ramp
        if len(busy) < 3:
            return None
        watts = sorted(w for w, _ in busy)
        mhz = sorted(m for _, m in busy if m)
        med = watts[len(watts) // 2]
        return {"package_w_median": round(med, 1), "package_w_max": round(watts[-1], 1), "cap_w": cap, "frac_of_cap": round(med / cap, 3) if cap else None,
                "package_w_at_start": round(idle, 1), "sclk_mhz_median": round(mhz[len(mhz) // 2], 1) if mhz else None, "samples": len(busy),
                "window_s": round(window, 3), "steps": steps,
                "source": f"hwmon of PCI function {bdf}: power1_input (PPT), power1_cap, freq1_input, read every 20 ms by a host thread during an extra "
                          "untimed run; the last two thirds of the window are used (the telemetry is a moving average)"}
    except Exception as ex:  # noqa: BLE001 -- a missing or unreadable sensor must never cost the bench line
        return {"error": str(ex)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=list(WORKLOADS), default="mul_relin_rescale")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="weak: --batch results PER GPU; strong: --batch is the GLOBAL batch.  Not given: N = 1 runs the batch once (the two are the same job); "
                         "N > 1 times BOTH in one invocation -- `value` = strong (north_star: batch 1024 over the GPUs of the node), a `weak` block beside it")
    ap.add_argument("--batch", type=int, default=0, help="operand-0 batch: per GPU (weak) or global (strong); 0: the configuration's own")
    ap.add_argument("--b1", type=int, default=0, help="operand-1 batch of the HEBench outer product (results = batch x b1; operand 1 is replicated on every rank); 0: the "
                                                      "configuration's own (1).  `--config eltwise_mul --batch 16 --b1 16` is the 16 x 16 shape of configs[1] (SURVEY.md 8d cfg2)")
    ap.add_argument("--chunk", type=int, default=0, help="ops per kernel sequence (0: library default)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="results in the CPU-baseline sample (-1: the configuration's own, 0: skip)")
    ap.add_argument("--cpu-passes", type=int, default=5, help="timed passes of the CPU baseline (median reported)")
    ap.add_argument("--parity-sample", type=int, default=-1, help="results every rank checks against the oracle (-1: the configuration's own, 0: skip)")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / sharding only: the ranks meet (gloo), every rank computes its shard of "
                                                           "the global batch, rank 0 prints them; no GPU is touched (what the CPU test of the N>1 path runs)")
    ap.add_argument("--profile-mode", action="store_true", help="only warm-up + timed steps on the GPU (no CPU baseline, no parity sample, no extra "
                                                                "single-stream step): what rocprofv3 traces and PMC passes should see")
    ap.add_argument("--force-dist", action="store_true", default=os.environ.get("HE355_BENCH_FORCE_DIST", "0") not in ("", "0"),
                    help="take the N > 1 branch at --gpus 1 too (also HE355_BENCH_FORCE_DIST=1): the one rank is started under torch.distributed.run, initialises the "
                         "collective backend (nccl = RCCL) on its device, and the line carries `collective` -- what a 1-GPU box can check of the multi-GPU path")
    args = ap.parse_args()
    if args.profile_mode:
        args.cpu_sample, args.parity_sample = 0, 0

    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # plain `python bench.py --gpus N`: this process becomes the launcher
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    use_dist = world > 1 or args.force_dist

    import torch
    import torch.distributed as dist
    if args.dry_run:
        raise SystemExit(dry_run(args, rank, world, dist, torch))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one rank per GPU; HE355_BENCH_BACKEND=gloo lets several ranks share one GPU for rehearsals on a 1-GPU box
    backend = os.environ.get("HE355_BENCH_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
    tdev = "cuda" if backend == "nccl" else "cpu"

    be = importlib.import_module("reference-seal-backend_amd")
    sharding = load_sharding()
    W = WORKLOADS[args.config]
    if args.b1 > 0:
        if args.config in ("bfv_matmul",):
            raise SystemExit("--b1 applies to the outer-product workloads only")
        W = type(W.__name__ + f"_b1_{args.b1}", (W,), {"b1": args.b1})
    bits = W.bits or be.chain_bits(W.depth, W.coeff_bits)
    ctx = be.Context(be.SCHEME_CKKS if W.scheme == "ckks" else be.SCHEME_BFV, W.N, bit_sizes=bits, plain_bits=W.plain_bits, device=device)
    if args.chunk:
        ctx.set_chunk(args.chunk)

    batch = args.batch or W.default_batch

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def timed_run(scaling):
        """One complete measurement under `scaling`: this rank's shard of the global batch resident in HBM, W warm-up steps, barrier + synchronize,
        K timed steps, barrier + synchronize, MAX over ranks.  Returns the workload (its buffers still resident) and the figures."""
        global_b0 = batch * world if scaling == "weak" else batch
        shard = sharding.shard_outer_product(global_b0, W.b1, world, rank)  # contiguous block of operand-0 rows; operand 1 and keys replicated
        first_buf = len(ctx._bufs)
        wl = W(be, ctx, shard, args)
        wl.setup()
        for _ in range(args.warmup):
            wl.step()
        barrier()
        ctx.timer_begin()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        gpu_ms = ctx.timer_end()  # HIP events on the stream the kernels run on
        barrier()
        elapsed = time.perf_counter() - t0
        my_ms = elapsed / args.steps * 1e3
        per_rank_ms = [round(my_ms, 3)]
        if use_dist:
            tm = torch.tensor([my_ms], dtype=torch.float64, device=tdev)
            allt = [torch.zeros_like(tm) for _ in range(world)]
            dist.all_gather(allt, tm)
            per_rank_ms = [round(float(x.item()), 3) for x in allt]
            t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        total_results = global_b0 * W.b1 * args.steps
        return {"wl": wl, "scaling": scaling, "global_b0": global_b0, "elapsed": elapsed, "gpu_ms": gpu_ms, "per_rank_ms": per_rank_ms,
                "total_results": total_results, "value": total_results / elapsed, "bufs": (first_buf, len(ctx._bufs))}

    def release(run):
        """hand a finished run's slabs back to the context's pool before the next run is set up"""
        lo, hi = run["bufs"]
        for b in ctx._bufs[lo:hi]:
            b.free()
        del ctx._bufs[lo:hi]
        run["wl"] = None

    # secondary runs first (their slabs are released), the primary last: the probes and checker legs below work on its resident shard
    plan = scalings_of(args, world)
    secondary = []
    for sc in plan[1:]:
        r = timed_run(sc)
        n_sec = r["wl"].n
        release(r)
        secondary.append({"scaling": sc, "value": round(r["value"], 2), "unit": "ciphertext-ops/sec", "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 3),
                          "per_rank_ms": r["per_rank_ms"], "global_batch": r["global_b0"] * W.b1, "batch_per_gpu_rank0": n_sec,
                          "batch": (f"{batch} operand-0 rows PER GPU x {world} GPUs" if sc == "weak" else f"global batch {batch} cut over {world} GPUs")})
    run = timed_run(plan[0])
    wl, n = run["wl"], run["wl"].n
    scaling_kind, global_b0, elapsed, gpu_ms, per_rank_ms, total_results, value = (run[k] for k in ("scaling", "global_b0", "elapsed", "gpu_ms", "per_rank_ms",
                                                                                                "total_results", "value"))
    # what a reader of the N > 1 line needs to see that the ranks really met over the collective backend and which card each one drove:
    # an all-reduce of one 1 per rank (must equal the world size), every rank's own time and its device's PCI bus id
    props = torch.cuda.get_device_properties(device)
    bus = (f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', -1):02x}:{getattr(props, 'pci_device_id', 0):02x}"
           if hasattr(props, "pci_bus_id") else None)
    collective = None
    per_rank_device = [{"rank": 0, "local_device": device, "pci_bus_id": bus, "name": props.name}]
    if use_dist:
        ones = torch.ones(1, dtype=torch.float64, device=tdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        objs = [None] * world
        dist.all_gather_object(objs, {"rank": rank, "local_device": device, "pci_bus_id": bus, "name": props.name})
        per_rank_device = objs
        collective = {"backend": dist.get_backend() + (" (RCCL)" if backend == "nccl" else ""), "world_size": dist.get_world_size(),
                      "allreduce_of_ones": float(ones.item()),
                      "note": "rendezvous, barriers, MAX-over-ranks and this all-reduce only: no collective inside the timed region (SURVEY.md 8e)"}

    # ---- shader clock held under this load: one extra untimed step (same schedule as the timed ones) with a one-wave probe beside it that
    # samples s_memtime against the 100 MHz counter for ~80 % of a step (he355_clock_probe_begin; MI355X_MICROARCH.md "DVFS give-back") ----
    clock = None
    if n > 0 and not args.profile_mode:
        step_us = elapsed / args.steps * 1e6
        ctx.clock_probe_begin(max(200, int(step_us * 0.8)))
        wl.step()
        ctx.sync()
        mhz, covered = ctx.clock_probe_end()
        clock = {"sustained_mhz": round(mhz, 1), "seconds_covered": round(covered, 6),
                 "source": "in-run: d(s_memtime) / d(s_memrealtime) x 100 MHz of one probe wave running beside one extra untimed step"}

    # ---- package power and core clock under this load: an extra untimed run of ~1.5 s while a host thread reads the device's hwmon files
    # (power1_input = PPT in microwatts, power1_cap, freq1_input = sclk) every 20 ms -- sysfs reads, no subprocess; rank 0 at N = 1 only ----
    power = None
    if n > 0 and not args.profile_mode and rank == 0 and world == 1:
        power = sample_power(torch.cuda.get_device_properties(device), lambda k: ([wl.step() for _ in range(k)], ctx.sync()), elapsed / args.steps)

    # ---- dominant kernel, single stream: one extra untimed step with the two-stream schedule off, HIP events around each launch ----
    k3 = None
    if args.config in ("mul_relin_rescale", "mul_relin", "dot") and n > 0 and not args.profile_mode:
        ctx.set_dual_stream(False)
        ctx.timer_begin()
        wl.step()
        ss_ms = ctx.timer_end()
        k3_ms, k3_launches, k3_ops = ctx.probe_dominant_kernel()
        ctx.set_dual_stream(True)
        k3 = (k3_ms, k3_launches, k3_ops, ss_ms)

    # ---- checker legs: every rank checks a sample of ITS shard; rank 0 at N=1 times the CPU baseline ----
    parity, cpu, checksum = None, None, None
    psample = W.parity_sample if args.parity_sample < 0 else args.parity_sample
    csample = W.default_cpu_sample if args.cpu_sample < 0 else args.cpu_sample
    do_cpu = rank == 0 and world == 1 and csample > 0
    rows_checked = 0
    if n > 0 and (do_cpu or psample > 0):
        import oracle as ho  # the checker: never inside the timed region, never the thing measured except as `cpu_baseline`
        o = ho.Context(ho.SCHEME_CKKS if W.scheme == "ckks" else ho.SCHEME_BFV, W.N, bit_sizes=bits, plain_bits=W.plain_bits)
        assert [int(q) for q in o.moduli] == [int(q) for q in ctx.moduli]
        cpus = ho.effective_cpus()  # affinity mask and cgroup quota: omp_get_max_threads() sees neither a quota nor a CPU share
        cores = max(1, min(ho.lib().ho_max_threads(), cpus["affinity"] or 1 << 30))
        threads = cores if world == 1 else max(1, min(cores, cpus["effective"] or cores) // world)  # the ranks of a node share its CPU quota
        res_wanted = csample if do_cpu else psample
        rows_checked = min(wl.rows, max(1, -(-res_wanted // W.b1)))  # whole operand-0 rows
        wl.step()  # (the extra single-stream step above left the same results; this keeps the legs independent of it)
        ctx.sync()
        if do_cpu:
            # The reference's loop is `omp parallel for collapse(2) num_threads(NumThreads)` over the batch (ckks eltwise .cpp:138-141,325).
            # Thread scaling of the stand-in on THIS host: 1 thread, a few counts, every hardware thread; `value` is the best of them.
            # Each count gets a sample of >= 2 results per thread sized for a few seconds per pass.
            counts = sorted({c for c in (1, 4, 16, 32, 64, cpus["effective"], threads) if 1 <= c <= threads})
            scaling, best = [], None
            per_thread_rate = None
            for tcount in counts:
                want_res = min(rows_checked * W.b1, max(8 if tcount == 1 else 2 * tcount, int((per_thread_rate or 4.0) * tcount * 2.5)))
                rows_t = min(rows_checked, max(1, -(-want_res // W.b1)))
                passes = args.cpu_passes if tcount == threads else 2
                tt, want_t = wl.checker(ho, o, rows_t, tcount, passes)
                med = sorted(tt)[len(tt) // 2]
                rate = rows_t * W.b1 / med
                if tcount == 1:
                    per_thread_rate = rate
                ok_t = bool(np.array_equal(wl.result_rows(rows_t * W.b1), want_t))
                scaling.append({"threads": tcount, "ops_per_sec": round(rate, 3), "results": rows_t * W.b1, "passes": len(tt), "seconds_per_pass": round(med, 3),
                                "efficiency_vs_1_thread": round(rate / (per_thread_rate * tcount), 3) if per_thread_rate else None, "parity": ok_t})
                if best is None or rate > best["ops_per_sec"]:
                    best = scaling[-1]
                if tcount == threads:
                    times, want, rows_done = tt, want_t, rows_t
            rows_checked = rows_done
            got = wl.result_rows(rows_checked * W.b1)
            parity = bool(np.array_equal(got, want)) and all(e["parity"] for e in scaling)
            cpu = {"value": best["ops_per_sec"], "unit": "ciphertext-ops/sec", "cores": best["threads"], "kind": "port",
                   "nproc": os.cpu_count(), "cpu_share": cpus, "scaling": scaling,
                   "one_thread_ops_per_sec": scaling[0]["ops_per_sec"] if scaling[0]["threads"] == 1 else None,
                   "built_for_host": "portable x86-64-v2 fallback (the in-tree -march=native build failed: oracle.build)" if ho.portable_build else
                   open(os.path.join(ROOT, "oracle", "_build", "host.sig")).read().strip()
                   if os.path.exists(os.path.join(ROOT, "oracle", "_build", "host.sig")) else None,
                   "sample": f"thread counts {counts}; per count a sample of the first rows of the resident batch (>= 2 results per thread, same parameters and "
                             f"keys as the GPU step), median of the timed passes after one warm-up; value = the best count ({best['threads']} threads, "
                             f"{best['results']} results, {best['seconds_per_pass']} s per pass); loop shape `omp parallel for` over the flattened (i, x) result index "
                             "as the reference's operate() (ckks eltwise .cpp:325), per-thread scratch as SEAL's thread-local pools (:343); in-repo "
                             "SEAL-algorithm restatement (oracle/he_oracle.c) built -O3 -march=native on this host; SEAL v3.7.2 unavailable offline"}
        else:
            times, want = wl.checker(ho, o, rows_checked, threads, 1)
            got = wl.result_rows(rows_checked * W.b1)
            parity = bool(np.array_equal(got, want))
    if n > 0:
        checksum = hashlib.sha256(np.ascontiguousarray(wl.result_rows(min(n, 4))).tobytes()).hexdigest()[:16]

    # every rank's verdict reaches rank 0
    flags = [1 if parity is None else int(parity), rows_checked * W.b1]
    if use_dist:
        t = torch.tensor(flags, dtype=torch.int64, device=tdev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [[int(v) for v in g.tolist()] for g in gathered]
    else:
        per_rank = [flags]
    all_ok = all(p[0] == 1 for p in per_rank)

    if rank == 0:
        bytes_op = wl.bytes_per_op(global_b0 * W.b1)
        achieved = bytes_op * total_results / elapsed / 1e9 / world  # per GPU: the roofline is one GPU's HBM
        # SURVEY.md 8d prices every result with both operands read; in a b0 x 1 outer product operand 1 is ONE ciphertext that stays in
        # cache, so the bytes that must cross HBM are fewer: operands read once per batch, results written once
        comp_op = wl.compulsory_bytes_per_op(global_b0, W.b1)
        # The newest committed counter file of THIS configuration (tools/profile_cfg.sh -> tools/kernel_bounds.py --json: PMC passes of one bench
        # step, each in its own rocprofv3 run): HBM bytes and VALU wave-instructions per result.  Not measured in this run -- PMC passes cannot
        # run inside it -- so the file is named next to every number taken from it.
        prof = committed_counters(args.config, W.b1, n)
        ops_per_gpu_s = total_results / elapsed / world
        traffic, traffic_src, valu = None, None, None
        hbm_traffic_frac = None
        if prof:
            pj, pfile = prof
            traffic = pj["hbm_bytes_per_op"] * n  # HBM bytes of one step's kernel sequence on one GPU (PMC: FETCH_SIZE x 2 + WRITE_SIZE)
            hbm_traffic_frac = pj["hbm_bytes_per_op"] * ops_per_gpu_s / HBM_PEAK
            traffic_src = {"file": pfile, "hbm_bytes_per_op": pj["hbm_bytes_per_op"], "ratio_to_algorithmic": round(pj["hbm_bytes_per_op"] / bytes_op, 2),
                           "hbm_GBps_of_real_bytes": round(pj["hbm_bytes_per_op"] * ops_per_gpu_s / 1e9, 1), "frac_of_8TBps": round(hbm_traffic_frac, 4),
                           "note": "PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 runs of one step of this command, tools/profile_cfg.sh): bytes per "
                                   "result from that file x the results of one step here; not measured in this run"}
            mhz = clock["sustained_mhz"] if clock and clock["sustained_mhz"] > 100 else pj.get("sustained_mhz_time_weighted")
            if mhz and pj.get("valu_wave_instr_per_op"):
                issue_peak = N_SIMD * mhz * 1e6 / 4  # a wave64 VALU instruction holds its 16-lane SIMD for 4 cycles
                winstr_s = pj["valu_wave_instr_per_op"] * ops_per_gpu_s
                valu = {"wave_instr_per_op": pj["valu_wave_instr_per_op"], "source": pfile + " (SQ_INSTS_VALU summed over the step's kernels / results)",
                        "sustained_mhz": mhz, "sustained_mhz_source": clock["source"] if clock and clock["sustained_mhz"] > 100 else pfile + " (GRBM_GUI_ACTIVE / 8 / kernel time)",
                        "issue_peak_wave_instr_per_s": round(issue_peak, 0), "achieved_wave_instr_per_s": round(winstr_s, 0),
                        "frac": round(winstr_s / issue_peak, 4),
                        "note": "1024 SIMDs x clock / 4 cycles per wave64 instruction; the fraction of VALU issue slots the step's kernels fill at the clock the chip "
                                "held in this run"}
        fr_valu = valu["frac"] if valu else None
        if fr_valu is not None and hbm_traffic_frac is not None:
            bound = "valu-issue" if fr_valu >= hbm_traffic_frac else "hbm"
            bound_why = (f"VALU issue {fr_valu:.3f} of peak vs HBM (real bytes) {hbm_traffic_frac:.3f} of 8 TB/s: the larger fraction names the binding resource")
        else:
            bound = "hbm"
            bound_why = "no committed counter file for this configuration: HBM is the roofline SURVEY.md 8d assigns it"
        roof = {"bound": bound, "bound_why": bound_why,
                "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 5),
                "traffic": traffic, "traffic_unit": "bytes per step per GPU" if traffic is not None else None, "traffic_source": traffic_src,
                "valu": valu, "clock_probe": clock, "power": power,
                "algorithmic_bytes_per_op": round(bytes_op, 1),
                "compulsory_bytes_per_op": round(comp_op, 1), "frac_of_compulsory": round(comp_op * total_results / elapsed / world / HBM_PEAK, 5),
                "kernel": "whole kernel sequence of the step (per GPU); the dominant kernel's own figures are under dominant_kernel",
                "gpu_ms_per_step_hip_events": round(gpu_ms / args.steps, 3),
                "note": "achieved / peak / frac: algorithmic bytes (SURVEY.md 8d) x results / time against 8 TB/s, per GPU.  `valu` prices the same run against the "
                        "VALU issue rate (64-bit modular butterflies: SURVEY.md 0.6, DESIGN.md 5); `bound` names the larger fraction"}
        if k3 is not None and k3[1] > 0:
            k3_ms, k3_launches, k3_ops, ss_ms = k3
            L, K, N = wl.L, wl.K, W.N
            n_f = sum(1 for i in list(range(L)) + [K - 1] if ctx.fp64[i])
            fused = os.environ.get("HE355_K3_FUSE", "1") != "0"
            # per key switch, in residue polynomials of the fp64-engine primes: (L - 1) lifted-digit rows (48-bit packed, 6 B per element)
            # + 1 own-digit row; fused floor steps: + 2 correction rows + 2 c01 rows in + 2 result rows out; unfused: 2 sums out
            polys8 = n_f * (1 + (6 if fused else 2))
            k3_bytes_ks = N * (6 * n_f * (L - 1) + 8 * polys8)
            roof["dominant_kernel"] = {
                "name": "k_k3<ArF64> (forward row pass of the lifted digits + key MAC + fused floor steps, fp64-engine primes)",
                "measured": "one extra untimed step with the two-stream schedule off (he355_set_dual_stream(0)): HIP events around every launch",
                "launches_per_step": k3_launches, "key_switches_per_step": k3_ops // max(1, (2 if args.config == "mul_relin_rescale" else 1)),
                "ms_per_step_single_stream": round(k3_ms, 3), "avg_launch_ms": round(k3_ms / k3_launches, 4),
                "step_ms_single_stream": round(ss_ms, 3),
                "algorithmic_bytes_per_key_switch": k3_bytes_ks,
                "note": "two launches per chunk with the fused rescale (the tiles of the prime divided out, then the rest)"}
        out = {
            "metric": "ciphertext-ops/sec" + (" (CKKS ct x ct mul+relin+rescale, N=2^15, L=16)" if args.config == "mul_relin_rescale" else f" ({args.config})"),
            "value": round(value, 2),
            "unit": "ciphertext-ops/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": scaling_kind,
            "vs_baseline": round(value / cpu["value"], 2) if cpu and cpu.get("value") else None,
            "vs_baseline_note": ("value / cpu_baseline.value: the in-repo CPU port of the reference's algorithm timed in this run on this box's cores -- NOT a published "
                                 "number (BASELINE.md holds none; SEAL itself is unavailable offline)") if cpu and cpu.get("value") else None,
            "dtype": "u64 (exact residues; fp64-FMA engine for primes < 2^47, u64 Harvey engine with the fold reduction for the 60-bit primes)",
            "data": "synthetic (uniform residues generated in HBM; synthetic evaluation keys; the same global batch at every world size)",
            "config": {"workload": wl.describe(), "poly_modulus_degree": W.N, "coeff_modulus_bits": bits,
                       "batch": (f"{global_b0} x {W.b1} results globally; scaling = {scaling_kind}: "
                                 + (f"{batch} operand-0 rows PER GPU x {world} GPU(s) -- per-GPU work is the batch `metric` names at every N, so the line "
                                    "answers `ciphertext-ops/sec at N GPUs` with each GPU as loaded as the 1-GPU run" if scaling_kind == "weak" else
                                    f"the GLOBAL batch is {batch} at every N ({n} results on rank 0) -- answers north_star's `batch = 1024 ... >= 6x at 8 GPUs` "
                                    "(the default at N > 1; the `weak` block beside it times the same batch PER GPU)")
                                 + "; rank r owns a contiguous block of operand-0 rows (sharding.shard_outer_product)"),
                       "batch_per_gpu": n, "global_batch": global_b0 * W.b1,
                       "parallelism": f"batch-sharded x{world}, keys and operand 1 replicated (built per device from the shared seed), no data-path collective"},
            "collective": collective,
            "weak": next((b for b in secondary if b["scaling"] == "weak"), None),
            "other_scalings": secondary or None,
            "per_rank_ms": per_rank_ms,
            "per_rank_device": per_rank_device,
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": {"checked_in_run": all_ok if any(p[1] for p in per_rank) else None,
                       "per_rank": [{"rank": r, "ok": bool(p[0]), "results_checked": p[1]} for r, p in enumerate(per_rank)],
                       "against": "oracle (CPU restatement) on the first rows of each rank's shard, bit for bit"},
            "results_checksum_rank0": checksum,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    ctx.close()
    if not all_ok:
        raise SystemExit("parity check failed: the HIP path's results differ from the oracle's (see parity.per_rank)")


if __name__ == "__main__":
    main()
