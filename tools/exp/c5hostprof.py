#!/usr/bin/env python3
"""cProfile of the host side of C5 steps (two channels, preprocessing, co-localisation): where the Python time goes.

    python tools/exp/c5hostprof.py [steps]
"""
import cProfile, functools, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from magellanmapper_amd import blob_log as bl, config, stack_detect, synth, _native as nat

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = bench.CONFIGS["c5"]
shape = cfg["shape"]
dev = torch.device("cuda", 0)
profile = dict(bench._BASE_PROFILE, **cfg["profile"])
config.resolutions = bench.RESOLUTIONS; config.filename = "p"
config.setup_roi_profiles(None); config.roi_profile.update(profile)
for p in config.roi_profiles:
    p.update(profile)
config.near_max = [-1.0, -1.0]
nat.keep_host_heap()
c0 = synth.make_volume_device(shape, 3, dev)
c1 = synth.make_volume_device(shape, 4, dev)
c1 = torch.maximum(c1.to(torch.int32), (c0.to(torch.int32) * 7) // 10).to(c0.dtype)
dvol = bl.DeviceVolume(torch.stack((c0, c1), dim=-1).contiguous())
del c0, c1
blocks = stack_detect.setup_blocks(config.roi_profile, shape)
bl.blob_log_blocks = functools.partial(bl.blob_log_blocks, budget_bytes=16 << 30)


def step():
    stack_detect.StackDetector.plan_pruning(blocks.overlap, blocks.tol, blocks.overlap_padding, [0, 1])
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(None, dvol, blocks.sub_roi_slices, blocks.sub_rois_offsets,
                                                           blocks.denoise_max_shape, None, True, [0, 1])
    return stack_detect.StackPruner.prune_blobs_mp(dvol, seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices,
                                                   blocks.sub_rois_offsets, [0, 1], blocks.overlap_padding,
                                                   final_form=True, untouched=True, n_flag_cols=2)


step(); step()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.1f} ms per step under the profiler")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
