#!/usr/bin/env python3
"""Measurement for the widened rows: a BASELINE.json configs[4]-style run --
2-channel light-sheet-like stack, stock preprocessing (denoise_size 25), isotropic rescale, intensity
co-localisation, optional spectral unmixing -- through StackDetector.detect_blobs_sub_rois + StackPruner,
with the per-kernel-family HIP-event timers on.  Prints one JSON line (kept under profiles/).

    python tools/rowbench.py [--shape Z Y X] [--res Z Y X] [--steps K] [--no-iso] [--no-coloc] [--unmix]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=3, default=[192, 1024, 1024])
ap.add_argument("--res", type=float, nargs=3, default=[3.0, 1.0, 1.0], help="voxel size z y x (um)")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--no-iso", action="store_true")
ap.add_argument("--no-coloc", action="store_true")
ap.add_argument("--no-denoise", action="store_true")
ap.add_argument("--unmix", action="store_true", help="channel 1 minus 0.2 x channel 0 before detection")
a = ap.parse_args()

from magellanmapper_amd import _native as nat, blob_log as bl, config, stack_detect, synth, detector

dev = torch.device("cuda", 0)
shape = tuple(a.shape)
config.setup_roi_profiles(None)
config.resolutions = np.array([a.res])
config.filename = "rowbench"
over = dict(segment_size=150, num_sigma=5, min_sigma_factor=2.6, max_sigma_factor=4.0,
            denoise_size=None if a.no_denoise else 25, isotropic=None if a.no_iso else (0.96, 1, 1))
config.roi_profile.update(over)
for p in config.roi_profiles:
    p.update(over)
if a.unmix:      # (not built together with the isotropic rescale: implies --no-iso)
    a.no_iso = True
    for p in config.roi_profiles:
        p["isotropic"] = None
        p.spectral_unmixing = {1: {0: 0.2}}
config.near_max = [-1.0, -1.0]

# two channels: the same blob field with 70 % of the blobs shared (co-localised), different seeds for the rest
t0 = time.time()
c0 = synth.make_volume_device(shape, 3, dev)
c1 = synth.make_volume_device(shape, 4, dev)
c1 = torch.maximum(c1.to(torch.int32), (c0.to(torch.int32) * 7) // 10).to(c0.dtype)    # (no uint16 max on the device)
vol = torch.stack((c0, c1), dim=-1).contiguous()
del c0, c1
torch.cuda.synchronize()
t_gen = time.time() - t0
dvol = bl.DeviceVolume(vol)
blocks = stack_detect.setup_blocks(config.roi_profile, shape)
n_blocks = int(np.prod(blocks.sub_roi_slices.shape))
chls = [0, 1]


def one_step():
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(
        None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets, blocks.denoise_max_shape,
        blocks.exclude_border, not a.no_coloc, chls)
    pruned, _ = stack_detect.StackPruner.prune_blobs_mp(
        dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets, chls,
        blocks.overlap_padding)
    return pruned


one_step()
nat.timing_enable(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    out = one_step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / a.steps
kt = nat.timing_read()
nat.timing_enable(False)
nvox = int(np.prod(shape))
iso = None if a.no_iso else [float(v) for v in np.asarray(a.res) / min(a.res) * np.array([0.96, 1, 1])]
res = {
    "tool": "tools/rowbench.py", "workload": f"{shape[2]}x{shape[1]}x{shape[0]} (x,y,z) x 2 channels uint16, "
    f"resolutions {a.res}, {n_blocks} blocks; denoise_size {over['denoise_size']}, isotropic factor {iso}, "
    f"coloc {not a.no_coloc}, unmix {bool(a.unmix)}, 5 sigmas; detect both channels + co-localise + prune",
    "ms_per_step": round(el * 1e3, 2),
    "Mvoxels_per_s_per_channel_pair": round(nvox / el / 1e6, 1),
    "Mvoxel_channels_per_s": round(2 * nvox / el / 1e6, 1),
    "blobs": 0 if out is None else int(len(out)),
    "kernels_ms_per_step": {k: round(ms / a.steps, 3) for k, (ms, n) in kt.items() if n},
    "launches_per_step": {k: n // a.steps for k, (ms, n) in kt.items() if n},
    "note": "'generic' = spectral unmixing + min/max + trilinear resize kernels (mmx_tables.hip); "
            "'preproc' = saturate/denoise tiles; 'coloc' = per-blob channel means",
    "volume_gen_s": round(t_gen, 2),
}
print(json.dumps(res))
