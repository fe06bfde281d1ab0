#!/usr/bin/env python3
"""Where the non-overlapped time of a step goes (wall-clock marks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from magellanmapper_amd import blob_log as bl, config, stack_detect, synth, _native as nat
import bench
shape = bench.SHAPE
dev = torch.device("cuda", 0)
config.resolutions = bench.RESOLUTIONS; config.filename = "p"
config.setup_roi_profiles(None); config.roi_profile.update(bench.PROFILE)
vol = synth.make_volume_device(shape, 3, dev)
dvol = bl.DeviceVolume(vol)
blocks = stack_detect.setup_blocks(config.roi_profile, shape)
import functools
bl.blob_log_blocks = functools.partial(bl.blob_log_blocks, budget_bytes=64 << 30)
marks = []
orig_finish = bl._finish_detect; orig_enq = bl._enqueue_detect; orig_prune = bl._prune_batch
def T(name, f):
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); marks.append((name, t, time.perf_counter())); return r
    return g
bl._finish_detect = T("finish", orig_finish); bl._enqueue_detect = T("enqueue", orig_enq); bl._prune_batch = T("prune_batch", orig_prune)
for rep in range(3):
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets, None, None, False, [0])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t1s = time.perf_counter()
    out = stack_detect.StackPruner.prune_blobs_mp(dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets, [0], blocks.overlap_padding)
    t2 = time.perf_counter()
    pruned = out[0]
    pruned[:, 0:3] = pruned[:, 7:10]
    t3 = time.perf_counter()
    final = pruned[:, [0, 1, 2, 3, 4, 5, 6, 10]]
    t4 = time.perf_counter()
print("final formatting (bench.py one_step): rel<-abs %.2f ms, column selection %.2f ms" % ((t3 - t2) * 1e3, (t4 - t3) * 1e3))
print("detect %.1f ms (gpu idle wait %.2f) prune %.1f ms total %.1f" % ((t1 - t0) * 1e3, (t1s - t1) * 1e3, (t2 - t1s) * 1e3, (t2 - t0) * 1e3))
for n, a, b in marks:
    print("  %-12s start %7.1f  dur %6.1f" % (n, (a - t0) * 1e3, (b - a) * 1e3))
