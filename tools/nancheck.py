#!/usr/bin/env python3
"""Every mode of mmx_log_batch_f32 on a NaN-filled workspace: a mode that leaves part of its output unwritten shows."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from magellanmapper_amd import _native as nat, blob_log as bl, kernels1d as k1, synth
L = nat.lib()
bad = 0
for shape, sigmas in (((48, 64, 72), (2.0, 3.0, 4.0, 5.0, 6.0)), ((70, 90, 150), (3.0, 4.0, 5.0)), ((30, 40, 50), (1.0, 2.0))):
    vol = synth.make_volume(5, shape, 40)
    dvol = bl.DeviceVolume(vol)
    blocks, slot = bl._make_blocks(dvol, 0, [(0, 0, 0)], [shape])
    d_blocks = bl._to_device_bytes(blocks, dvol.tensor.device)
    v32 = dvol.view(0, True)
    for mode in (0, 2, 3, 4, 5, 6, 7, -1):
        ns = len(sigmas)
        ws = torch.full(((4 + ns) * slot,), float("nan"), dtype=torch.float32, device=dvol.tensor.device)
        path = ctypes.c_int(0)
        paths = []
        for i, s in enumerate(sigmas):
            R = k1.kernel_radius(s)
            w0, w2 = k1.gaussian_half_kernel(s, 0, R), k1.gaussian_half_kernel(s, 2, R)
            nat.check(L.mmx_log_batch_f32(ctypes.byref(v32), d_blocks.data_ptr(), blocks.ctypes.data, 1, slot,
                                          nat.as_double_ptr(w0), nat.as_double_ptr(w2), R, s * s,
                                          ws.data_ptr() + (4 + i) * slot * 4, ws.data_ptr(), None, 0.0, 0.0, None, mode,
                                          ctypes.byref(path), torch.cuda.current_stream().cuda_stream), "log")
            paths.append(path.value)
        torch.cuda.synchronize()
        px = int(blocks["px"][0])
        cube = ws[4 * slot:].view(ns, slot)[:, :shape[0] * shape[1] * px].view(ns, shape[0], shape[1], px)[..., :shape[2]].cpu().numpy()
        nan = [int(np.isnan(cube[i]).sum()) for i in range(ns)]
        ok = not any(nan)
        bad += 0 if ok else 1
        print(("ok  " if ok else "FAIL"), shape, "mode", mode, "paths", paths, "NaNs per sigma", nan)
print("failures:", bad)
