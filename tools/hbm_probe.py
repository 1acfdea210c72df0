#!/usr/bin/env python3
"""HBM streaming rates of one MI355X as torch sees them (write-only, read-only, copy): the ceilings the HBM-bound kernels
(k_k1, k_k2, k_floor_rows) are compared with in DESIGN.md."""
import torch

dev = torch.device("cuda", 0)
n = 1 << 30  # 8 GiB of float64
a = torch.empty(n, dtype=torch.float64, device=dev)
b = torch.empty(n, dtype=torch.float64, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


ms = timed(lambda: a.fill_(1.0))
print(f"write-only (fill 8 GiB): {8 * 2**30 / ms / 1e9:.2f} TB/s")
ms = timed(lambda: torch.sum(a))
print(f"read-only (sum 8 GiB): {8 * 2**30 / ms / 1e9:.2f} TB/s")
ms = timed(lambda: b.copy_(a))
print(f"copy (8 GiB -> 8 GiB): {16 * 2**30 / ms / 1e9:.2f} TB/s total traffic")
ms = timed(lambda: torch.add(a, b, out=b))
print(f"add (2 reads + 1 write): {24 * 2**30 / ms / 1e9:.2f} TB/s total traffic")
