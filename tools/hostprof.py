#!/usr/bin/env python3
"""cProfile of one full-volume step (host-side overheads)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from magellanmapper_amd import blob_log as bl, config, stack_detect, synth
import bench
shape = bench.SHAPE if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1:4])
dev = torch.device("cuda", 0)
config.resolutions = bench.RESOLUTIONS; config.filename = "p"
config.setup_roi_profiles(None); config.roi_profile.update(bench.PROFILE)
vol = synth.make_volume_device(shape, 3, dev)
dvol = bl.DeviceVolume(vol)
blocks = stack_detect.setup_blocks(config.roi_profile, shape)
def step():
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets, None, None, False, [0])
    return stack_detect.StackPruner.prune_blobs_mp(dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets, [0], blocks.overlap_padding)
step()
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
