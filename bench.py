#!/usr/bin/env python3
"""Headline benchmark: CKKS multiply -> relinearize -> rescale, N=2^15, L=16, batch 1024 per GPU.

Metric (BASELINE.json): ciphertext-ops/sec, one op = one result ciphertext of the per-pair pipeline
(multiply, relinearize_inplace, rescale_to_next_inplace — the sequence at
/root/reference/src/benchmarks/ckks/seal_ckks_matmultval_benchmark.cpp:253-255 applied per batch element with
the HEBench outer-product indexing of seal_ckks_element_wise_benchmark.cpp:322-336, here 1024 x 1).

A "step" is one pass of that pipeline over the whole resident batch.  Inputs (uniform residues — a uniformly
random ciphertext is distribution-identical to a real one) and the relinearization key are
generated in HBM before the timed region.  One process per GPU; the batch is sharded by replication of the
workload (weak scaling: 1024 results per GPU), no data-path collective: the units are independent
(SURVEY.md §8e).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N = 32768
DEPTH = 16
COEFF_BITS = 45
BATCH = 1024
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md


def algorithmic_bytes_per_op(L: int, n: int, batch: int, K: int) -> float:
    """read 2 cts + write 1 ct at L-1 + the relin key once per batch."""
    ct_in = 2 * L * n * 8
    ct_out = 2 * (L - 1) * n * 8
    key = L * 2 * K * n * 8
    return 2 * ct_in + ct_out + key / batch


def cpu_baseline(be, sample_ops: int, bits):
    """Oracle (CPU port of the SEAL-algorithm pipeline) timed on this host's cores on a bounded sample;
    the same sample is pushed through the GPU path and compared bit-for-bit."""
    import oracle as ho
    o = ho.Context(ho.SCHEME_CKKS, N, bit_sizes=bits)
    rng = np.random.default_rng(1234)
    L = o.L
    a = np.stack([o.random_poly(rng, L, 2) for _ in range(sample_ops)])
    b = o.random_poly(rng, L, 2)[None]
    rk = o.random_kswitch_key(rng)
    threads = ho.lib().ho_max_threads()
    idx_a = np.arange(sample_ops, dtype=np.uint32)
    idx_b = np.zeros(sample_ops, dtype=np.uint32)
    o.batch_op(ho.OP_MUL_RELIN_RESCALE, a[:threads], idx_a[:threads], b, idx_b[:threads], rk)  # warm-up (tables, pages)
    times = []
    for _ in range(3):  # median of three timed passes over the sample (after the warm-up above)
        t0 = time.perf_counter()
        want = o.batch_op(ho.OP_MUL_RELIN_RESCALE, a, idx_a, b, idx_b, rk)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    return dict(value=sample_ops / dt, seconds=dt, cores=threads, a=a, b=b, rk=rk, want=want, total_seconds=sum(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH, help="results per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="ops per kernel sequence (0: library default)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="ops in the CPU-baseline sample (0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one rank per GPU; HE355_BENCH_BACKEND=gloo lets several ranks share one GPU for rehearsals on a 1-GPU box
    backend = os.environ.get("HE355_BENCH_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)

    be = importlib.import_module("reference-seal-backend_amd")
    bits = be.chain_bits(DEPTH, COEFF_BITS)
    ctx = be.Context(be.SCHEME_CKKS, N, bit_sizes=bits, device=device)
    L, K = ctx.L, ctx.K
    n = args.batch
    if args.chunk:
        ctx.set_chunk(args.chunk)

    # ---- resident inputs (HBM) ----
    pm = list(range(L))
    d_a = ctx.alloc(n * 2 * L * N)
    d_b = ctx.alloc(1 * 2 * L * N)
    d_out = ctx.alloc(n * 2 * (L - 1) * N)
    ctx.fill_uniform(d_a, n * 2 * L, pm, 1234 + rank)
    ctx.fill_uniform(d_b, 2 * L, pm, 99 + rank)
    ctx.set_relin_key_synthetic(7)
    ix = be.Context.outer(0, n, 0, 1)

    def step():
        ctx.multiply_relin(L, n, d_a, d_b, ix, d_out, rescale=True)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gpu_ms = ctx.timer_end()  # HIP events on the stream the kernels run on
    k3_ms, k3_launches, k3_ops = ctx.probe_dominant_kernel()  # HIP events around every k_k3<fp64> launch of the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_ops = n * args.steps * world
    value = total_ops / elapsed

    cpu = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:  # the CPU baseline is a single-GPU-run item (rank 0 at N=1 only)
        cb = cpu_baseline(be, args.cpu_sample, bits)
        # push the identical sample through the GPU path: the measured path is the checked path
        s = args.cpu_sample
        da, db = ctx.to_device(cb["a"]), ctx.to_device(cb["b"])
        ctx.set_relin_key(cb["rk"])
        do = ctx.alloc(s * 2 * (L - 1) * N)
        ctx.multiply_relin(L, s, da, db, be.Context.outer(0, s, 0, 1), do, rescale=True)
        parity = bool(np.array_equal(do.download(cb["want"].shape), cb["want"]))
        cpu = {"value": round(cb["value"], 3), "unit": "ciphertext-ops/sec", "cores": cb["cores"], "kind": "port",
               "sample": f"{s} of the {n} results of one step (same parameters, uniform residues), "
                         f"median of 3 passes, {cb['seconds']:.2f} s wall each ({cb['total_seconds']:.1f} s in all) on {cb['cores']} OpenMP threads; "
                         "in-repo SEAL-algorithm restatement, SEAL v3.7.2 unavailable offline"}

    if rank == 0:
        bytes_op = algorithmic_bytes_per_op(L, N, n, K)
        gpu_s = gpu_ms / 1e3
        # HBM bytes per op from the PMC passes of the same command (tools/profile_round.sh -> profiles/r01_hbm_traffic.json:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath) and n == BATCH:
            traffic = json.load(open(tpath))["hbm_bytes_per_op"] * n
        achieved = bytes_op * n * args.steps / gpu_s / 1e9
        # dominant kernel: k_k3 for the fp64-engine primes (key products of the key switch, with both floor steps finished in
        # its epilogue by default).  Algorithmic bytes per op of that kernel, in residue polynomials:
        # n_f * (L - 1) lifted-digit rows in (48-bit packed: 6 B per element) + n_f own-digit rows in; fused: + 2 n_f correction rows (one combined correction per residue; the
        # prime divided out has its mod-down correction) + 2 n_f c01 rows in + 2 (n_f - 1) result rows + 2 c01 rows out;
        # unfused: 2 n_f sums out.  Plus the key rows of those primes once per chunk.
        n_f = sum(1 for i in list(range(L)) + [K - 1] if ctx.fp64[i])
        fused = os.environ.get("HE355_K3_FUSE", "1") != "0"
        k3_polys = n_f * (L + 6) if fused else n_f * (L + 2)
        k3_bytes_op = N * (6 * n_f * (L - 1) + 8 * (k3_polys - n_f * (L - 1)))
        k3_key_bytes = L * 2 * n_f * N * 8 * (n * args.steps / float(args.chunk or 256))  # once per chunk of ops
        k3_bytes_total = k3_bytes_op * n * args.steps + k3_key_bytes
        k3_gbps = k3_bytes_total / (k3_ms / 1e3) / 1e9 if k3_ms > 0 else None
        out = {
            "metric": "ciphertext-ops/sec (CKKS ct x ct mul+relin+rescale, N=2^15, L=16)",
            "value": round(value, 2),
            "unit": "ciphertext-ops/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (exact fp64-FMA engine for the 45-bit primes, u64 Harvey for the 60-bit primes)",
            "data": "synthetic (uniform residues generated in HBM; synthetic relinearization key)",
            "config": {"workload": "CKKS EltwiseMult + relinearize + rescale, N=2^15, depth 16 (L=16 data primes + 1 special), "
                                   f"batch {n}x1 per GPU (BASELINE.json configs[2])",
                       "poly_modulus_degree": N, "coeff_modulus_bits": bits, "batch_per_gpu": n, "global_batch": n * world,
                       "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved * 1e9 / HBM_PEAK, 5), "traffic": traffic,
                         "traffic_note": "HBM bytes per step (1024 ops) from PMC counters; algorithmic bytes per step = %d" % int(bytes_op * n),
                         "kernel": "mul->relin->rescale kernel sequence per chunk (k_k1, k_k2, k_k3 special prime, k_floor_cols, k_k3 data primes with fused mod-down, k_floor_cols, k_floor_rows)",
                         "algorithmic_bytes_per_op": bytes_op,
                         "dominant_kernel": {"name": "k_k3<ArF64> (forward row pass of the lifted digits + key MAC + fused floor steps, fp64-engine primes)",
                                             "launches": k3_launches, "total_ms_hip_events": round(k3_ms, 3),
                                             "avg_launch_ms_hip_events": round(k3_ms / max(1, k3_launches), 4),
                                             "ms_per_step": round(k3_ms / args.steps, 3),
                                             "share_of_gpu_time": round(k3_ms / gpu_ms, 3) if gpu_ms else None,
                                             "algorithmic_bytes_per_op": k3_bytes_op,
                                             "achieved_GBps": round(k3_gbps, 1) if k3_gbps else None,
                                             "frac_of_hbm_peak": round(k3_gbps * 1e9 / HBM_PEAK, 4) if k3_gbps else None,
                                             "note": "two launches per chunk (the tiles of the prime the rescale divides out, then the rest); durations "
                                                     "overlap the other stream's kernels (two-stream schedule), as in the rocprofv3 kernel trace of the same command"},
                         "gpu_ms_per_step_hip_events": round(gpu_ms / args.steps, 3),
                         "note": "expected binding resource is the VALU (64-bit modular butterflies), not HBM: SURVEY.md §0.6"},
            "cpu_baseline": cpu,
            "parity_checked_in_run": parity,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
