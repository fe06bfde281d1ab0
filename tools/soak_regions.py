#!/usr/bin/env python3
"""Randomised check (host only, no GPU): the region-wise pruning (stack_detect._RegionPruner, as blocks land) against
the whole-table passes on synthetic block tables -- random stack shapes, block sizes, voxel sizes (tolerances),
border exclusion (wider overlaps, no padding), tolerance factors, channels, duplicate jitter.

    python tools/soak_regions.py [n_trials] [seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from magellanmapper_amd import config, stack_detect as sd
from test_host_logic import _synthetic_block_tables

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
done = skipped = 0
for trial in range(n_trials):
    res = rng.choice([0.5, 0.8, 1.0, 1.6, 2.5], 3)
    config.resolutions = np.array([res])
    config.setup_roi_profiles(None)
    seg = int(rng.integers(20, 70))
    eb = None if rng.random() < 0.6 else tuple(int(v) for v in rng.integers(0, 4, 3))
    ptf = (1, 1, 1) if rng.random() < 0.5 else tuple(float(v) for v in rng.choice([1, 0.9, 0.6, 1.5], 3))
    config.roi_profile.update(segment_size=seg, denoise_size=None, exclude_border=eb, prune_tol_factor=ptf)
    shape = tuple(int(v) for v in rng.integers(30, 200, 3))
    blocks = sd.setup_blocks(config.roi_profile, shape)
    if not all(sd.StackPruner._axis_geometry(a, shape, blocks.overlap, blocks.overlap_padding, blocks.sub_roi_slices,
                                             blocks.sub_rois_offsets)[1]
               for a in range(3) if blocks.sub_rois_offsets.shape[a] > 1) or np.prod(blocks.sub_roi_slices.shape) > 400:
        skipped += 1
        continue
    channels = [0] if rng.random() < 0.7 else [0, 1]
    n_extra = 0 if rng.random() < 0.7 else 2
    tables = _synthetic_block_tables(rng, shape, blocks, int(rng.integers(50, 6000)), channels,
                                     jitter=int(rng.integers(0, 9)), n_extra=n_extra)
    grid = blocks.sub_roi_slices.shape
    coords = list(np.ndindex(*grid))
    share = list(range(len(coords)))

    def build(with_pruner):
        arena = sd._TableArena(11 + n_extra, len(share))
        pruner = None
        if with_pruner:
            plan = sd.StackPruner._axis_plan(shape, blocks.overlap, blocks.tol, blocks.overlap_padding,
                                             blocks.sub_roi_slices, blocks.sub_rois_offsets)
            pruner = sd._RegionPruner(arena, plan, channels, blocks.sub_roi_slices, shape, share)
        segr = np.zeros(grid, dtype=object).view(sd._SegRois)
        for k in share:
            tbl = tables[coords[k]]
            if tbl is not None:
                arena.add(coords[k], tbl)
                tbl = arena.view(coords[k])
            arena.landed()
            segr[coords[k]] = tbl
            if pruner is not None:
                pruner.advance()
        for k in share:          # (views taken once everything has landed, as assemble_seg_rois does: the arena stays "intact")
            if segr[coords[k]] is not None:
                segr[coords[k]] = arena.view(coords[k])
        assert arena.intact(segr)
        segr.arena, segr.pruner = arena, pruner
        return segr

    class Img:
        pass
    Img.shape = shape
    args = (blocks.overlap, blocks.tol, blocks.sub_roi_slices, blocks.sub_rois_offsets, channels, blocks.overlap_padding)
    if not any(t is not None for t in tables.values()):
        skipped += 1
        continue
    want, df_want = sd.StackPruner.prune_blobs_mp(Img, build(False), *args)
    seg_b = build(True)
    pruner_b = seg_b.pruner                 # (prune_blobs_mp takes it off the tables: one shot)
    got, df_got = sd.StackPruner.prune_blobs_mp(Img, seg_b, *args)
    assert all(d is not None for d in pruner_b.done), "the regions were not merged"
    ok = np.array_equal(got, want) and np.array_equal(df_got.to_numpy(), df_want.to_numpy())
    done += 1
    if not ok:
        print(f"MISMATCH trial {trial}: shape {shape} res {res} seg {seg} eb {eb} ptf {ptf} channels {channels} "
              f"rows {len(want)} vs {len(got)}")
        sys.exit(1)
print(f"region-wise pruning == whole-table passes in {done} trials ({skipped} skipped: irregular geometry / empty)")
