#!/usr/bin/env python3
"""Plain HBM rates of this box for reference: fill (write only), read-reduce (read only), copy (both)."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 30   # floats: 4 GiB
a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.zero_()); print(f"fill   {4 * n / ms / 1e6:8.1f} GB/s")
ms = t(lambda: a.fill_(1.5)); print(f"fill   {4 * n / ms / 1e6:8.1f} GB/s")
ms = t(lambda: a.sum()); print(f"read   {4 * n / ms / 1e6:8.1f} GB/s")
ms = t(lambda: b.copy_(a)); print(f"copy   {8 * n / ms / 1e6:8.1f} GB/s (read + write)")
ms = t(lambda: torch.add(a, 1.0, out=b)); print(f"add    {8 * n / ms / 1e6:8.1f} GB/s (read + write)")
