#!/usr/bin/env python3
"""Known-byte streaming launches for calibrating FETCH_SIZE / WRITE_SIZE (run under rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magellanmapper_amd import _native as nat
dev = torch.device("cuda", 0)
n = 1 << 30     # 1 Gi elements: 4 GiB float / 2 GiB u16, far beyond the 256 MiB Infinity Cache
a = torch.rand(n, device=dev)
b = torch.empty_like(a)
u = torch.zeros(n, dtype=torch.int16, device=dev)
L = nat.lib()
s = torch.cuda.current_stream().cuda_stream
for kind, src in ((0, a), (1, a), (2, u)):
    for _ in range(2):
        nat.check(L.mmx_calib_stream(kind, src.data_ptr(), b.data_ptr(), n, s), "calib")
torch.cuda.synchronize()
print("calib done: n =", n)
