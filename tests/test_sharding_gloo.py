"""N > 1 path on the CPU: two gloo ranks shard an outer-product batch with the very call bench.py makes
(sharding.shard_outer_product on the global batch: batch x world for weak scaling, batch for strong), exchange their shard
descriptors and timings the way bench.py does (barrier, MAX over ranks, all_gather of the per-rank parity flags), and check
that the shards tile the result range exactly and that the operand-0 fill offsets (he355_fill_uniform_at first_poly) tile the
global operand array."""
import importlib.util
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_sharding():
    spec = importlib.util.spec_from_file_location("he355_sharding", os.path.join(ROOT, "reference-seal-backend_amd", "sharding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["he355_sharding"] = mod
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, b0, b1, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _load_sharding().shard_outer_product(b0, b1, world, rank, value_index0=3)
    mine = torch.tensor([sh.first_result, sh.n_results, sh.a_base, sh.a_count], dtype=torch.int64)
    gathered = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, mine)
    dist.barrier()
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)  # pretend rank r took 10*(r+1) ms
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put(([g.tolist() for g in gathered], float(t.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("b0,b1", [(1024, 1), (7, 3), (1, 5)])
def test_two_rank_sharding(b0, b1):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + b0 % 7
    procs = [ctx.Process(target=_worker, args=(r, world, port, b0, b1, q)) for r in range(world)]
    for p in procs:
        p.start()
    shards, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(tmax - 0.020) < 1e-12  # MAX over ranks
    covered = []
    for first, n, a_base, a_count in shards:
        assert n == a_count * b1 and first == (a_base - 3) * b1
        covered += list(range(first, first + n))
    assert covered == list(range(b0 * b1))  # every result exactly once, in order, no overlap
    mod = _load_sharding()
    assert mod.aggregate_throughput([s[1] for s in shards], [0.01, 0.02]) == pytest.approx(b0 * b1 / 0.02)


def test_bench_shards_are_world_size_independent():
    """bench.py: rank r fills rows [a_base, a_base + a_count) of the GLOBAL operand-0 array (first_poly = a_base * 2 * L), so the union
    over ranks is the same array at every world size; strong scaling keeps the global batch, weak scaling multiplies it."""
    mod = _load_sharding()
    L = 16
    for world in (1, 2, 4, 8):
        for scaling, batch in (("weak", 1024), ("strong", 1024), ("strong", 1001)):
            g = batch * world if scaling == "weak" else batch
            shards = [mod.shard_outer_product(g, 1, world, r) for r in range(world)]
            polys = []
            for sh in shards:
                first = sh.a_base * 2 * L
                polys += list(range(first, first + sh.a_count * 2 * L))
            assert polys == list(range(g * 2 * L))
            assert sum(sh.n_results for sh in shards) == g
            assert max(sh.n_results for sh in shards) - min(sh.n_results for sh in shards) <= 1  # balanced to one row


def _bench(*argv, env=None, timeout=180):
    import json
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):  # as the driver calls it: no launcher around it
        e.pop(k, None)
    e.update(env or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run around it (how the driver runs N=1): the process starts its own two
    ranks, they rendezvous (gloo), every rank computes its shard of the global batch, rank 0 prints one JSON line."""
    for config, scaling, batch, b0 in (("mul_relin_rescale", "strong", 7, 7), ("mul_relin_rescale", "weak", 5, 10), ("dot", "strong", 9, 9), ("dot", "weak", 4, 8)):
        p, doc = _bench("--gpus", "2", "--dry-run", "--config", config, "--scaling", scaling, "--batch", str(batch), env={"HE355_BENCH_BACKEND": "gloo"})
        assert p.returncode == 0, p.stderr[-2000:]
        assert doc["dry_run"] and doc["n_gpus"] == 2 and doc["global_b0"] == b0 and doc["config"] == config
        want = [_load_sharding().shard_outer_product(b0, 1, 2, r) for r in range(2)]
        assert [(s["rank"], s["a_base"], s["a_count"], s["n_results"]) for s in doc["shards"]] == [(w.rank, w.a_base, w.a_count, w.n_results) for w in want]
        assert sum(s["a_count"] for s in doc["shards"]) == b0  # the shards tile the global batch


def test_bench_times_both_scalings_at_n_gt_1_by_default():
    """`python bench.py --gpus N` with no --scaling (how the driver runs it): N > 1 plans BOTH scalings in the one invocation -- the primary
    (`value`) is strong at the global batch `metric` names (north_star: batch 1024 over the GPUs), a weak run (1024 per GPU) beside it --
    and N = 1 plans one run.  --force-dist takes the distributed branch with one rank."""
    p, doc = _bench("--gpus", "2", "--dry-run", env={"HE355_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert [r["scaling"] for r in doc["runs"]] == ["strong", "weak"] and doc["scaling"] == "strong"
    assert doc["runs"][0]["global_b0"] == 1024 and [s["a_count"] for s in doc["runs"][0]["shards"]] == [512, 512]
    assert doc["runs"][1]["global_b0"] == 2048 and [s["a_count"] for s in doc["runs"][1]["shards"]] == [1024, 1024]
    assert [s["a_base"] for s in doc["runs"][1]["shards"]] == [0, 1024]
    p, doc = _bench("--gpus", "1", "--dry-run")
    assert p.returncode == 0 and [r["scaling"] for r in doc["runs"]] == ["weak"] and doc["global_b0"] == 1024
    p, doc = _bench("--gpus", "1", "--dry-run", "--force-dist", env={"HE355_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert doc["n_gpus"] == 1 and doc["shards"][0]["a_count"] == 1024


def test_bench_self_launch_propagates_failure():
    """Without --dry-run the ranks need a GPU: on a box without one every rank exits loudly (no CPU fallback) and the launcher
    returns non-zero; on a GPU box two gloo ranks share the card and the run succeeds."""
    import torch
    p, doc = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--cpu-sample", "0", "--parity-sample", "1",
                    env={"HE355_BENCH_BACKEND": "gloo"}, timeout=600)
    if torch.cuda.is_available():
        assert p.returncode == 0, p.stderr[-2000:]
        assert doc["n_gpus"] == 2 and doc["parity"]["checked_in_run"] is True
    else:
        assert p.returncode != 0 and doc is None
        assert "no HIP device" in p.stderr or "needs an MI355X" in p.stderr


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu_through_the_self_launch_path():
    """The N>1 compute path on the one-GPU box: `python bench.py --gpus 2` (no launcher), two gloo ranks on the same card, each on its
    own shard of the global batch with its own parity sample against the oracle."""
    p, doc = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "6", "--scaling", "strong", "--cpu-sample", "0",
                    "--parity-sample", "1", env={"HE355_BENCH_BACKEND": "gloo"}, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert doc["n_gpus"] == 2 and doc["config"]["global_batch"] == 6 and doc["scaling"] == "strong"
    assert doc["parity"]["checked_in_run"] is True and [r["ok"] for r in doc["parity"]["per_rank"]] == [True, True]
    assert doc["collective"]["world_size"] == 2 and doc["collective"]["allreduce_of_ones"] == 2.0 and len(doc["per_rank_ms"]) == 2


@pytest.mark.gpu
def test_bench_both_scalings_in_one_invocation_on_the_shared_gpu():
    """The default N > 1 invocation on the one-GPU box (two gloo ranks share the card): the line's `value` is the strong run, the `weak`
    block carries its own value / ms_per_step / per_rank_ms, and the CPU-thread list of the baseline never lands in `scaling`."""
    p, doc = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "6", "--cpu-sample", "0", "--parity-sample", "1",
                    env={"HE355_BENCH_BACKEND": "gloo"}, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert doc["scaling"] == "strong" and doc["config"]["global_batch"] == 6 and doc["config"]["batch_per_gpu"] == 3
    w = doc["weak"]
    assert w["scaling"] == "weak" and w["global_batch"] == 12 and w["batch_per_gpu_rank0"] == 6 and len(w["per_rank_ms"]) == 2 and w["value"] > 0
    assert doc["parity"]["checked_in_run"] is True


def test_oracle_build_is_locked_and_host_stamped(tmp_path):
    """oracle.build(): N fresh ranks may call it at once (bench.py on a box whose `_build/` did not travel) -- a file lock lets exactly
    one run make; the library carries the signature of the host it was built on (`-march=native`) and is rebuilt on another host."""
    import subprocess
    import sys
    import oracle
    if os.environ.get("HE_ORACLE_LIB_PATH"):
        pytest.skip("HE_ORACLE_LIB_PATH names the library (tools/asan_oracle.sh): oracle.build() does not build")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sig = os.path.join(root, "oracle", "_build", "host.sig")
    oracle.build()
    assert open(sig).read().strip() == oracle._host_signature()
    # a stale signature (the tree came from another host) triggers exactly one rebuild however many processes ask at once
    open(sig, "w").write("some other host\n")
    code = "import oracle, os; oracle.lib(); print(os.path.getmtime(oracle._LIB_PATH))"
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, text=True) for _ in range(3)]
    outs = [p.communicate(timeout=300)[0].strip() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert len(set(outs)) == 1, outs                       # all three loaded the same, single rebuild
    assert open(sig).read().strip() == oracle._host_signature()
    cpus = oracle.effective_cpus()
    assert 1 <= cpus["effective"] <= (cpus["nproc"] or 1)
