#!/usr/bin/env python3
"""Preprocessing micro-benchmark: HIP-event time of mmx_preprocess_batch on a batch of blocks.

    python tools/ppbench.py [--blocks 27] [--edge 266] [--dms 25] [--reps 3] [--generic]
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magellanmapper_amd import _native as nat, blob_log as bl, synth, config, preprocess

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=27)
ap.add_argument("--edge", type=int, default=266)
ap.add_argument("--dms", type=int, nargs="+", default=[25])
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--generic", action="store_true")
ap.add_argument("--no-unsharp", action="store_true")
ap.add_argument("--no-erosion", action="store_true")
ap.add_argument("--mode", choices=("auto", "single", "pipelined"), default="auto")
ap.add_argument("--tpw", type=int, default=0, help="tiles per workgroup of the pipelined blur kernel")
ap.add_argument("--profile", action="store_true", help="phase ticks of a -DPP_PROFILE build (MMX_LIB_PATH)")
ap.add_argument("--check", action="store_true", help="compare the float64 output with the single-kernel form")
a = ap.parse_args()
dev = torch.device("cuda", 0)
e = a.edge
g = int(np.ceil(a.blocks ** (1 / 3)))
step = e - 10
shape = (step * g + 10,) * 3
vol = synth.make_volume_device(shape, 3, dev)
dvol = bl.DeviceVolume(vol)
origins = [(z * step, y * step, x * step) for z in range(g) for y in range(g) for x in range(g)][:a.blocks]
shapes = [(e, e, e)] * len(origins)
config.setup_roi_profiles(None)
if a.no_unsharp:
    config.roi_profile["unsharp_strength"] = 0
if a.no_erosion:
    config.roi_profile["erosion_threshold"] = 0
dms = a.dms * 3 if len(a.dms) == 1 else a.dms
preprocess.FORCE_GENERIC = a.generic
preprocess.KERNEL_MODE = dict(auto=nat.MMX_PP_AUTO, single=nat.MMX_PP_SINGLE, pipelined=nat.MMX_PP_PIPELINED)[a.mode]
preprocess.TILES_PER_WG = a.tpw
pre = preprocess.Preprocessor(dms, want_info=True)
nvox = len(origins) * e ** 3
times = []
for rep in range(a.reps + 1):
    if rep == 1:
        nat.timing_enable(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    pre.run(dvol, 0, origins, shapes, 0)
    t1.record(); t1.synchronize()
    if rep >= 1:
        t = nat.timing_read()
        times.append((t["preproc"][0], t0.elapsed_time(t1)))
info = pre.info()
if a.profile:
    import ctypes
    buf = (ctypes.c_ulonglong * 32)()
    nat.lib().mmx_pp_profile_read(buf, 1)
    t = np.array(buf[:], dtype=np.float64).reshape(2, 16) / 100.0 / len(info) / (a.reps + 1)       # us per tile
    print("stats us/tile: load+hist %.2f select %.2f hist2 %.2f select2 %.2f tally+info %.2f" % tuple(t[0, :5]))
    print("blur  us/tile: first load %.2f setup %.2f z %.2f y %.2f x %.2f unsharp(last part) %.2f erode+end %.2f reload %.2f | issue prefetch %.2f batches %.2f" % tuple(t[1, :10]))
if a.check:
    got = [t.copy() for t in pre.fetch(shapes)]
    got_info = info.copy()
    preprocess.KERNEL_MODE = nat.MMX_PP_SINGLE
    ref = preprocess.Preprocessor(dms, want_info=True)
    ref.run(dvol, 0, origins, shapes, 0)
    want = ref.fetch(shapes)
    bad = sum(int((g != w).sum()) for g, w in zip(got, want))
    wi = ref.info()
    print("check: %d differing voxels of %d; flags equal %s; vmin/vmax equal %s; max |mean diff| %.3g" % (
        bad, sum(g.size for g in got), bool((wi["flags"] == got_info["flags"]).all()),
        bool((wi["vmin"] == got_info["vmin"]).all() and (wi["vmax"] == got_info["vmax"]).all()),
        float(np.abs(wi["mean"] - got_info["mean"]).max())))
if os.environ.get("PP_PROFILE"):
    a1 = np.floor(info["vmin"]); a2 = (info["vmin"] - a1) * 1e6
    b1 = np.floor(info["vmax"]); b2 = (info["vmax"] - b1) * 1e6
    print("stage ticks (100 MHz) load %.0f select %.0f saturate %.0f blur %.0f write %.0f" % (
        a1.mean(), a2.mean(), b1.mean(), b2.mean(), info["mean"].mean()))
k = float(np.mean([t[0] for t in times])); w = float(np.mean([t[1] for t in times]))
print(json.dumps(dict(blocks=len(origins), edge=e, dms=dms, tiles=len(info), kernel_ms=round(k, 3),
                      wall_ms=round(w, 3), gvox_per_s=round(nvox / k / 1e6, 2),
                      us_per_tile_per_cu=round(k * 1e3 / (len(info) / 256), 2),
                      eroded=int((info["flags"] & 2 != 0).sum()), identity=int((info["flags"] & 1 != 0).sum()))))
