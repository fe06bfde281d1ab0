#!/usr/bin/env python3
"""Median host time of each part of a small stack's step (c2), 300 steps, no profiler.

    python tools/exp/c2parts.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from magellanmapper_amd import blob_log as bl, config, detector, stack_detect, synth, _native as nat

cfg = bench.CONFIGS["c2"]
shape = cfg["shape"]
dev = torch.device("cuda", 0)
config.resolutions = bench.RESOLUTIONS; config.filename = "p"
config.setup_roi_profiles(None); config.roi_profile.update(dict(bench._BASE_PROFILE, **cfg["profile"]))
vol = synth.make_volume_device(shape, cfg["seed"], dev)
dvol = bl.DeviceVolume(vol)
blocks = stack_detect.setup_blocks(config.roi_profile, shape)
acc = {}


def wrap(obj, name, tag):
    fn = getattr(obj, name)
    raw = fn.__func__ if hasattr(fn, "__func__") and isinstance(obj, type) and isinstance(obj.__dict__.get(name), classmethod) else None

    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return (raw(obj, *a, **k) if raw is not None else fn(*a, **k))
        finally:
            acc.setdefault(tag, []).append(time.perf_counter() - t)
    if raw is not None:
        setattr(obj, name, inner)
    elif isinstance(obj, type) and isinstance(obj.__dict__.get(name), staticmethod):
        setattr(obj, name, staticmethod(inner))
    else:
        setattr(obj, name, inner)


wrap(bl, "_enqueue_detect", "enqueue")
wrap(bl._NativeEvent, "synchronize", "  wait for the batch's event")
wrap(bl, "_finish_detect", "finish (wait + resolve)")
wrap(bl, "_resolve_peaks_native", "  resolve (native + numpy around it)")
wrap(bl, "_prune_batch_native", "overlap prune per block")
wrap(stack_detect._ArenaSink, "__call__", "tables -> arena")
wrap(stack_detect.StackPruner, "_prune_table", "  three axis passes")
wrap(stack_detect.StackPruner, "_take_rows", "  gather of the output table")
wrap(stack_detect.StackPruner, "_ratio_frame", "  ratio frame")
wrap(stack_detect.StackPruner, "_ratios_from_counts", "  ratios from counts")
wrap(stack_detect.StackPruner, "_geometry", "  geometry memo")
wrap(stack_detect.StackDetector, "assemble_seg_rois", "  assemble_seg_rois")
wrap(bl, "blob_log_blocks", "  blob_log_blocks")
wrap(detector, "detect_blobs_blocks_device", " detect_blobs_blocks_device")


class _Timed:
    def __init__(self, fn, tag):
        self.fn, self.tag = fn, tag

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        acc.setdefault(self.tag, []).append(time.perf_counter() - t)
        return r


_L = nat.lib()
for _n in ("mmx_host_resolve_peaks", "mmx_host_overlap_prune", "mmx_host_emit_tables", "mmx_host_prune_region",
           "mmx_host_prune_axis", "mmx_host_take_rows", "mmx_host_map_columns", "mmx_detect_batch", "mmx_graph_launch",
           "mmx_event_synchronize"):
    if hasattr(_L, _n):
        setattr(_L, _n, _Timed(getattr(_L, _n), "    native " + _n))


def step():
    t0 = time.perf_counter()
    stack_detect.StackDetector.plan_pruning(blocks.overlap, blocks.tol, blocks.overlap_padding, [0])
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets,
                                                           None, None, False, [0])
    t1 = time.perf_counter()
    pruned, _ = stack_detect.StackPruner.prune_blobs_mp(dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                        blocks.sub_rois_offsets, [0], blocks.overlap_padding,
                                                        final_form=True, untouched=True)
    t2 = time.perf_counter()
    if isinstance(pruned, stack_detect._FinalTable):
        detector.Blobs(None).cols = list(pruned.col_names); final = pruned.view(np.ndarray)
    else:
        bb = detector.Blobs(pruned); bb.replace_rel_with_abs_blob_coords(pruned); final = bb.remove_abs_blob_coords(True)
    t3 = time.perf_counter()
    acc.setdefault("detect_blobs_sub_rois", []).append(t1 - t0)
    acc.setdefault("prune_blobs_mp", []).append(t2 - t1)
    acc.setdefault("final columns", []).append(t3 - t2)
    acc.setdefault("STEP", []).append(t3 - t0)
    return final


for _ in range(30):
    step()
acc.clear()
for _ in range(300):
    step()
for k, v in acc.items():
    v = np.asarray(v) * 1e3
    print(f"{k:42s} median {np.median(v):6.3f}  min {v.min():6.3f}  calls/step {len(v) / 300:.0f}")
