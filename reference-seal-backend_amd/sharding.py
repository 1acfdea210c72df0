"""Batch sharding across the GPUs of one node.

Every result r = i*b1 + x of an operate() call depends only on operand0[i], operand1[x] and the read-only keys
(/root/reference/src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:334-336), so the flattened result range is
cut into contiguous blocks, one per rank, with NO data-path collective: each rank runs the same kernel sequence on
its block.  Keys and (when b1 > 1) operand 1 are replicated; operand 0 is sharded by i.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    first_result: int  # first flattened result index owned by this rank
    n_results: int
    a_base: int        # value_index into operand 0 for this rank's first row
    a_count: int       # rows of operand 0 this rank needs
    b_base: int
    b_count: int       # operand 1 is replicated in full when b1 > 1


def shard_outer_product(b0: int, b1: int, world_size: int, rank: int, value_index0: int = 0, value_index1: int = 0) -> Shard:
    """Rank's share of a b0 x b1 outer product, cut along operand 0 so that no rank needs another rank's rows."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(b0, world_size)
    rows = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return Shard(rank=rank, first_result=row0 * b1, n_results=rows * b1, a_base=value_index0 + row0, a_count=rows,
                 b_base=value_index1, b_count=b1)


def aggregate_throughput(units_per_rank: list[int], seconds_per_rank: list[float]) -> float:
    """Whole-job rate as bench.py reports it: all units / the slowest rank's time."""
    return sum(units_per_rank) / max(seconds_per_rank)
