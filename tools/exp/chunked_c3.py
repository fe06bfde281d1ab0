#!/usr/bin/env python3
"""The benchmark volume (C3, 8.6 GB) detected from HOST memory resident as a whole and z-chunk by z-chunk
(stack_detect.MAX_RESIDENT_BYTES = 3 / 5 GB: one / two block layers per chunk): same table, and what the chunks cost.

    python tools/exp/chunked_c3.py
"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from magellanmapper_amd import blob_log as bl, config, stack_detect, synth

dev = torch.device("cuda", 0)
shape, seed = bench.CONFIGS["c3"]["shape"], bench.CONFIGS["c3"]["seed"]
vol = synth.make_volume_device(shape, seed, dev).cpu().numpy()
config.resolutions = bench.RESOLUTIONS
config.filename = "chunked"
config.setup_roi_profiles(None)
config.roi_profile.update(dict(bench._BASE_PROFILE, **bench.CONFIGS["c3"]["profile"]))
for p in config.roi_profiles:
    p.update(config.roi_profile)
config.near_max = [-1.0]
sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def run(limit):
    stack_detect.MAX_RESIDENT_BYTES = limit
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        _, _, blobs = stack_detect.detect_blobs_blocks("chunked", stack_detect.Image5d(vol[None]), None, None, [0],
                                                       False, False, True, False)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    return blobs.blobs, times


ref = None
for name, limit in (("resident", 1 << 40), ("chunks of <= 2.5 GB", 5 << 30), ("chunks of <= 1.5 GB", 3 << 30)):
    tbl, times = run(limit)
    ref = tbl if ref is None else ref
    print(f"{name:22s} {len(tbl)} blobs, sha1 {sha(tbl)[:12]}, same as resident: {np.array_equal(tbl, ref)}, "
          f"ms per call {' '.join('%.1f' % t for t in times)}, peak device memory "
          f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    torch.cuda.reset_peak_memory_stats()
