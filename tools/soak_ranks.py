#!/usr/bin/env python3
"""Randomised soak of the multi-rank path (blocks sharded over ranks, table all-gather, rank-0 prune) against the
oracle: run under torch.distributed.run with any number of ranks; all ranks may share one GPU (gloo).

    MMX_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 \\
        --master-port 29533 tools/soak_ranks.py [--trials N] [--seed S]
"""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as tdist

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=10)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
if world > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    tdist.init_process_group(os.environ.get("MMX_DIST_BACKEND", "gloo"))
from magellanmapper_amd import blob_log as bl, config, preprocess, stack_detect, synth
from oracle import magmap_oracle as mmo

preprocess.RGB_GUESS = True
rng = np.random.default_rng(a.seed)          # the same draws on every rank
bad = rows = 0
t0 = time.time()


def lexsorted(t):
    return t[np.lexsort(tuple(t[:, i] for i in range(t.shape[1] - 1, -1, -1)))]


for trial in range(a.trials):
    nch = int(rng.choice([1, 2]))
    shape = (int(rng.integers(30, 80)), int(rng.integers(60, 150)), int(rng.integers(60, 150)))
    chans = [synth.make_volume(int(rng.integers(1 << 30)), shape, int(rng.integers(20, 250)),
                               blob_sigma=float(rng.uniform(1.5, 3.5)), amp=float(rng.uniform(8000, 50000)))
             for _ in range(nch)]
    vol = chans[0] if nch == 1 else np.stack(chans, axis=-1)
    res = np.array([[float(rng.choice([1.0, 2.0, 3.0])), 1.0, 1.0]])
    coloc = bool(nch == 2 and rng.random() < 0.6)
    excl = None if rng.random() < 0.6 else tuple(int(v) for v in rng.integers(0, 6, 3))
    over = dict(segment_size=int(rng.choice([30, 44, 60])), num_sigma=int(rng.integers(2, 5)),
                detection_threshold=float(rng.choice([0.05, 0.1, 0.2])), overlap=float(rng.choice([0.3, 0.5, 0.8])),
                denoise_size=None if rng.random() < 0.6 else 25, isotropic=None, exclude_border=excl,
                prune_tol_factor=tuple(float(v) for v in rng.choice([0.5, 1.0, 1.5], 3)))
    config.setup_roi_profiles(["default"] * nch)
    for p in config.roi_profiles:
        p.update(over)
    config.roi_profile.update(over)
    config.resolutions = res
    config.filename = "soak"
    config.near_max = [-1.0] * nch
    chls = list(range(nch))
    blocks = stack_detect.setup_blocks(config.roi_profile, shape)
    seg = stack_detect.StackDetector.detect_blobs_sub_rois(
        None, vol, blocks.sub_roi_slices, blocks.sub_rois_offsets, blocks.denoise_max_shape, blocks.exclude_border,
        coloc, chls)
    if rank == 0:
        got, _ = stack_detect.StackPruner.prune_blobs_mp(
            bl.DeviceVolume(vol), seg, blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets,
            chls, blocks.overlap_padding)
        want = mmo.detect_blobs_blocks(vol, None, [dict(p) for p in config.roi_profiles], res,
                                       near_max=config.near_max, coloc=coloc)[1]["pruned11"]
        if got is not None:
            got[:, 0:3] = got[:, 7:10]          # as the oracle's stage table (rel <- abs, stack_detect.py:461)
        ok = (want is None and got is None) or (want is not None and got is not None and got.shape == want.shape and
                                                np.array_equal(lexsorted(got), lexsorted(want)))
        rows += 0 if want is None else len(want)
        if not ok:
            bad += 1
            print("MISMATCH trial", trial, shape, nch, res.tolist(), over, coloc,
                  None if got is None else got.shape, None if want is None else want.shape, flush=True)
    if world > 1:
        tdist.barrier()
if rank == 0:
    print(f"rank soak seed {a.seed}: {world} ranks, {a.trials} trials, {rows} pruned rows compared (incl. co-localisation "
          f"columns), {bad} mismatching stacks, {time.time() - t0:.0f} s")
if world > 1:
    tdist.destroy_process_group()
sys.exit(1 if bad else 0)
