"""Block geometry of a stack (mirror of the parts of ``magmap.cv.chunking`` on the path).

* :func:`stack_splitter` -- reference magmap/cv/chunking.py:214-256 (with ``_num_units``
  :170-185 and ``_bounds_side`` :188-211): the grid is ``ceil(shape / max_pixels)`` and
  block ``k`` spans ``[k * mp, min(k * mp + mp + overlap, size))`` along each axis.
* :func:`merge_blobs` -- :410-445: concatenate per-block tables, tagging each row with its
  block's grid coordinate.
* the multiprocessing helpers (:105-167) are kept as thin shims: blocks are dispatched to
  GPUs, not to a process pool, but callers may still query / set the start method.
"""
from __future__ import annotations

import multiprocessing as mp
from typing import Optional, Sequence, Tuple

import numpy as np

from . import config


def set_mp_start_method(val: Optional[str] = None) -> str:
    if val is None:
        val = config.roi_profile["mp_start"] if config.roi_profile else "fork"
    avail = mp.get_all_start_methods()
    if val not in avail:
        val = avail[0]
    try:
        mp.set_start_method(val)
    except RuntimeError:
        pass
    return val


def is_fork() -> bool:
    return mp.get_start_method(False) == "fork"


def get_mp_pool(initializer=None, initargs=None):
    """A ``multiprocessing.Pool`` sized by ``config.cpus`` and the first ROI profile's ``mp_max_tasks``
    (reference magmap/cv/chunking.py:143-167).  The detection path here sends blocks to the GPU, not to a pool;
    callers of the reference's helper (the match-based co-localiser, user scripts) still get their pool.  Worker
    processes must not touch the GPU the parent holds: use it for host-only work."""
    prof = config.get_roi_profile(0)
    max_tasks = None if not prof else prof["mp_max_tasks"]
    return mp.Pool(processes=config.cpus, maxtasksperchild=max_tasks, initializer=initializer,
                   initargs=() if initargs is None else initargs)


def stack_splitter(shape: Sequence[int], max_pixels: Sequence[int],
                   overlap: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """``(sub_roi_slices, sub_rois_offsets)``: an object array of slice triples indexed by
    block (z, y, x) and a float array ``grid + (3,)`` of block origins."""
    size = np.asarray(shape[:3])
    mp_ = np.asarray(max_pixels)
    grid = (-(-size // mp_)).astype(int)          # ceil division, per axis
    ov = np.zeros(3, dtype=int) if overlap is None else np.asarray(overlap)
    starts = [np.arange(g) * int(m) for g, m in zip(grid, mp_)]
    stops = [np.minimum(st + int(m) + int(o), int(n))
             for st, m, o, n in zip(starts, mp_, ov, size)]
    slices = np.zeros(grid, dtype=object)
    offsets = np.zeros(tuple(grid) + (3,))
    for coord in np.ndindex(*grid):
        slices[coord] = tuple(slice(int(starts[a][coord[a]]), int(stops[a][coord[a]]))
                              for a in range(3))
        offsets[coord] = [starts[a][coord[a]] for a in range(3)]
    return slices, offsets


def merge_blobs(blob_rois: np.ndarray) -> Optional[np.ndarray]:
    """All block tables stacked, with the block's grid coordinate as 3 extra int columns."""
    arena = getattr(blob_rois, "arena", None)
    if arena is not None and arena.n and arena.intact(blob_rois):
        # the tables were stored back to back (grid order) with their tags while the GPU was busy
        return arena.store[:arena.n]
    # (no rows at all: blocks whose blobs were all excluded hold EMPTY tables, not None, and the reference
    # then merges to an empty table rather than None -- the loop below does the same)
    live = [(coord, blob_rois[coord]) for coord in np.ndindex(*blob_rois.shape)
            if blob_rois[coord] is not None and not isinstance(blob_rois[coord], (int, np.integer))]
    if not live:
        return None
    n_rows = sum(t.shape[0] for _, t in live)
    n_cols = live[0][1].shape[1]
    out = np.empty((n_rows, n_cols + 3), dtype=np.result_type(live[0][1].dtype, np.int64))
    at = 0
    for coord, tbl in live:                       # one allocation, filled block by block
        out[at:at + tbl.shape[0], :n_cols] = tbl
        out[at:at + tbl.shape[0], n_cols:] = coord
        at += tbl.shape[0]
    return out
