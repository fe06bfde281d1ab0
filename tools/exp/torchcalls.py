#!/usr/bin/env python3
"""What the small torch calls on the host path cost (microseconds, median of 2000)."""
import time, numpy as np, torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
s2 = torch.cuda.Stream()
def t(name, fn, n=2000):
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    print(f"{name:46s} {np.median(ts) * 1e6:8.2f} us")
t("torch.cuda.is_available()", torch.cuda.is_available)
t("torch.cuda.current_device()", torch.cuda.current_device)
t("torch.cuda.current_stream()", torch.cuda.current_stream)
t("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
t("torch.device('cuda', current_device())", lambda: torch.device("cuda", torch.cuda.current_device()))
t("s2.wait_stream(current_stream())", lambda: s2.wait_stream(torch.cuda.current_stream()))
t("torch.cuda.Event() + record", lambda: torch.cuda.Event().record())
def _with():
    with torch.cuda.stream(s2):
        pass
t("with torch.cuda.stream(s2): pass", _with)
t("x.data_ptr()", x.data_ptr)
t("tuple(x.stride()), tuple(x.shape), str(x.dtype)", lambda: (tuple(x.stride()), tuple(x.shape), str(x.dtype)))
