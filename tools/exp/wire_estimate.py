#!/usr/bin/env python3
"""What the two exchanges of the distributed pruning cost a rank on the HOST side, measured on one GPU: a ONE-rank RCCL
group runs dist._gather_rows' device branch (pinned staging -> device -> all_gather_into_tensor -> device -> pinned,
stream synchronise) and dist.all_reduce_sum with the payload sizes of a C3 step at N ranks; the copy back is made N times
as large as one rank's rows (what a real gather returns).  No inter-GPU transfer in it: add bytes / link rate.

    python tools/exp/wire_estimate.py [N]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.distributed as td
from magellanmapper_amd import dist

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=dev, init_method="tcp://127.0.0.1:29611")
dist._force_collectives = True
rows_total, seam_frac = 330000, 0.10
own = rows_total // N
seam = np.random.rand(int(own * seam_frac * (2 if N > 2 else 1)), 10)          # rows near the neighbours' blocks
surv = np.random.rand(int(own * 0.885), 10)                                    # own survivors, final columns + key


def exchange(rows, width):
    """dist._gather_rows on one rank, with the copy back N times as large (the other ranks' rows)."""
    t0 = time.perf_counter()
    bufs, counts, w = dist._gather_rows(rows, width)
    most = len(rows)
    recv = torch.empty((N * most, width), dtype=torch.float64, device=dev)
    back = dist._pinned("recv_all", N * most * width)
    back[:N * most * width].view(N * most, width).copy_(recv, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(5):
    exchange(seam, 10); exchange(surv, 10); dist.all_reduce_sum(np.arange(3))
t = {"exchange 1 (seam rows)": [], "exchange 2 (survivors)": [], "all_reduce x 2": []}
for _ in range(50):
    t["exchange 1 (seam rows)"].append(exchange(seam, 10))
    t["exchange 2 (survivors)"].append(exchange(surv, 10))
    t0 = time.perf_counter()
    dist.all_reduce_sum(np.arange(3)); dist.all_reduce_sum(np.arange(40))
    t["all_reduce x 2"].append((time.perf_counter() - t0) * 1e3)
tot = 0.0
for k, v in t.items():
    print(f"N = {N}: {k:26s} median {np.median(v):6.3f} ms  (p95 {np.percentile(v, 95):6.3f})")
    tot += float(np.median(v))
mb = (len(seam) + len(surv)) * 10 * 8 * (N - 1) / 1e6
print(f"N = {N}: host side of the collectives per step ~ {tot:.2f} ms; bytes received per rank {mb:.1f} MB "
      f"(xGMI ring at ~100 GB/s: {mb / 100:.2f} ms)")
td.destroy_process_group()
