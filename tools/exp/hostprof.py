import sys, os, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch
from magellanmapper_amd import blob_log as bl, config, stack_detect, synth, _native as nat
import bench, functools
shape = bench.CONFIGS["c3"]["shape"]
dev = torch.device("cuda", 0)
config.resolutions = bench.RESOLUTIONS; config.filename = "p"
config.setup_roi_profiles(None); config.roi_profile.update(dict(bench._BASE_PROFILE, **bench.CONFIGS["c3"]["profile"]))
vol = synth.make_volume_device(shape, 3, dev)
dvol = bl.DeviceVolume(vol)
blocks = stack_detect.setup_blocks(config.roi_profile, shape)
bl.blob_log_blocks = functools.partial(bl.blob_log_blocks, budget_bytes=16 << 30)
def step():
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets, None, None, False, [0])
    out = stack_detect.StackPruner.prune_blobs_mp(dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets, [0], blocks.overlap_padding)
    return out
step(); step()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
