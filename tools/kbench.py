#!/usr/bin/env python3
"""Per-kernel micro-benchmark: HIP-event time of each kernel family on a batch of blocks.

    python tools/kbench.py [--blocks 32] [--edge 261] [--sigmas 3 4 5] [--reps 3]
"""
import argparse, ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magellanmapper_amd import _native as nat, blob_log as bl, synth

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=32)
ap.add_argument("--edge", type=int, default=261)
ap.add_argument("--sigmas", type=float, nargs="+", default=[3.0, 4.0, 5.0])
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--generic", action="store_true")
ap.add_argument("--mask", action="store_true", help="Y pass writes the NMS pre-filter masks; sparse NMS kernel")
a = ap.parse_args()
dev = torch.device("cuda", 0)
e = a.edge
# a volume holding `blocks` blocks of edge e, overlapping by 5 like the real grid
g = int(np.ceil(a.blocks ** (1 / 3)))
step = e - 5
shape = (step * g + 5, step * g + 5, -(-(step * g + 5) // 64) * 64)   # rows of whole 128-byte lines
vol = synth.make_volume_device(shape, 3, dev)
dvol = bl.DeviceVolume(vol)
origins = [(z * step, y * step, x * step) for z in range(g) for y in range(g) for x in range(g)][:a.blocks]
shapes = [(e, e, e)] * len(origins)
L = nat.lib()
blocks, slot = bl._make_blocks(dvol, 0, origins, shapes)
nb = len(blocks)
ns = len(a.sigmas)
ws = torch.empty((4 + ns) * nb * slot, dtype=torch.float32, device=dev)
d_blocks = bl._to_device_bytes(blocks, dev)
v32 = dvol.view(0, True)
stream = torch.cuda.current_stream().cuda_stream
from magellanmapper_amd import kernels1d as k1
nvox = nb * e ** 3   # algorithmic voxels (pitch columns not counted)
res = {}
fn = L.mmx_log_batch_f32_generic if a.generic else L.mmx_log_batch_f32
log_base = ws.data_ptr() + 4 * nb * slot * 4
mask_words = (nb * slot) >> 5
masks = torch.zeros(ns * mask_words * 2, dtype=torch.int64, device=dev)
written = ctypes.c_int(0)
zxp = ctypes.c_int(0)
ZX = int(os.environ.get('MMX_FUSE', -1))
PRE = os.environ.get('MMX_PREPACK', '1') == '1'
for rep in range(a.reps + 1):
    if rep == 1:
        nat.timing_enable(True)
    packed = False
    if ZX in (-1, 6, 7) and PRE and not a.generic:
        packed = L.mmx_zx_pack(ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot, ws.data_ptr(), stream) == 0
    for i, s in enumerate(a.sigmas):
        R = k1.kernel_radius(s)
        w0 = k1.gaussian_half_kernel(s, 0, R); w2 = k1.gaussian_half_kernel(s, 2, R)
        nat.check(fn(ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot,
                     nat.as_double_ptr(w0), nat.as_double_ptr(w2), R, s * s,
                     log_base + i * nb * slot * 4, ws.data_ptr(), *(() if a.generic else ((masks.data_ptr() + i * mask_words * 16) if a.mask else None, 0.1 - 2e-5, 2e-5,
                                                    ctypes.byref(written), ((ZX if ZX in (6, 7) else 6) | 0x100 | (0x200 if os.environ.get('MMX_Y_VALU') == '1' else 0)) if packed else ZX, ctypes.byref(zxp))), stream), "log")
        assert not a.mask or written.value in (1, 2)
        layout = written.value
        if rep >= 1:
            t = nat.timing_read()
            for k, (ms, n) in t.items():
                if n:
                    res.setdefault((k, R), []).append(ms)
    cap = 1 << 20
    table = torch.empty(cap * 48, dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    nat.check(L.mmx_peaks_batch(log_base, masks.data_ptr() if a.mask else None, layout, ns, d_blocks.data_ptr(), blocks.ctypes.data, nb, slot, 0.1, 2e-5,
                                table.data_ptr(), cap, count.data_ptr(), stream), "peaks")
    if rep >= 1:
        t = nat.timing_read()
        res.setdefault(("peaks", ns), []).append(t["peaks"][0])
torch.cuda.synchronize()
alg = {"zpass": 10, "ypass": 16, "xpass": 12, "generic": 38 / 3, "zxpass": 10, "y2pass": 12, "zxpack": 4}
print(f"blocks {nb} x {e}^3 = {nvox/1e6:.0f} Mvox; candidates {int(count.item())}; zx path {zxp.value}")
for (k, R), v in sorted(res.items()):
    ms = float(np.median(v))
    if k == "peaks":
        gbs = 4 * R * nvox / ms / 1e6
    else:
        gbs = alg.get(k, 0) * nvox / ms / 1e6
    print(f"{k:8s} R={R:3d}  {ms:8.3f} ms  {gbs:8.1f} GB/s(alg)  {nvox/ms/1e6:7.2f} Gvox/s")

if os.environ.get("ZX2_PROFILE"):
    # per (block, y) row: [producer busy, first consumer wave busy, total, last consumer wave busy] in 100 MHz ticks
    R = k1.kernel_radius(a.sigmas[-1])
    w0 = k1.gaussian_half_kernel(a.sigmas[-1], 0, R); w2 = k1.gaussian_half_kernel(a.sigmas[-1], 2, R)
    nat.check(fn(ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, nb, slot, nat.as_double_ptr(w0),
                 nat.as_double_ptr(w2), R, 1.0, log_base, ws.data_ptr(), None, 0.0, 0.0, None, ZX, None, stream), "log")
    torch.cuda.synchronize()
    P = ws[:nb * slot].view(nb, slot).cpu().numpy()
    px = int(blocks["px"][0])
    simd = P[:, :e * px].reshape(nb, e, px)[:, :, 8:24].reshape(-1, 16)
    import collections
    print("wave -> SIMD maps seen:", collections.Counter(tuple(int(v) for v in r[:14]) for r in simd).most_common(4))
    rows = P[:, :e * px].reshape(nb, e, px)[:, :, :5].reshape(-1, 5)
    print("R=%d ticks per row-WG (100 MHz): producer busy %.0f, tail producer busy %.0f, consumer busy %.0f / %.0f, total %.0f" % (
        R, rows[:, 0].mean(), rows[:, 4].mean(), rows[:, 1].mean(), rows[:, 3].mean(), rows[:, 2].mean()))
