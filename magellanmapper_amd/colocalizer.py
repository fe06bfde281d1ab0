"""Intensity co-localisation of blobs across channels (mirror of ``magmap.cv.colocalizer``'s
``colocalize_blobs``, reference magmap/cv/colocalizer.py:340-441).

A blob of channel A "co-localises" with channel B when the mean intensity of B over the voxels
the blob owns (a ``ball(2)`` around its centre; voxels contested by several blobs of the same
channel go to the higher table row) reaches B's threshold -- the smallest such mean among B's
own blobs.  The per-blob means come from the device (``mmx_coloc_means``, bit-equal float64 in
NumPy's summation order); thresholds and flags are a handful of NumPy reductions on the
``(n_blobs, n_channels)`` matrix.

Only the default ``thresh="min"`` of the block path (``StackDetector.detect_sub_roi``,
stack_detect.py:159-162) is built; a percentile threshold raises ``NotImplementedError``.
"""
from __future__ import annotations

import ctypes
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np

try:
    import torch
except ImportError:  # pragma: no cover
    torch = None

from . import _native as nat


def _flags_from_means(table: np.ndarray, means: np.ndarray, shape3, n_channels: int) -> np.ndarray:
    """Thresholds + flags of one block: ``means[b, c]`` = mean of channel ``c`` over blob ``b``'s
    voxels (NaN when it owns none).  Rows outside the ROI get zeros (colocalizer.py:375-378, 434-436)."""
    colocs = np.zeros((table.shape[0], n_channels), dtype=np.uint8)
    if table.shape[0] == 0:
        return colocs
    in_roi = np.all([table[:, 0] >= 0, table[:, 0] < shape3[0], table[:, 1] >= 0, table[:, 1] < shape3[1],
                     table[:, 2] >= 0, table[:, 2] < shape3[2]], axis=0)
    chl = table[:, 6]
    present = np.unique(chl[in_roi]).astype(int)
    for other in present:
        if other < 0 or other >= n_channels:
            raise IndexError(f"index {other} is out of bounds for axis 0 with size {n_channels}")
        own = in_roi & (chl == other)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            thr = np.amin(means[own, other])          # NaN (a blob that owns nothing) poisons it
        hit = in_roi & np.isin(chl, present) & (means[:, other] >= thr)
        colocs[hit, other] = 1
    return colocs


def colocalize_blocks_device(volumes: Dict[int, nat.Volume], blocks: np.ndarray, d_blocks,
                             shapes, tables: List[Optional[np.ndarray]], n_channels: int,
                             dev, means_only: bool = False) -> List[Optional[np.ndarray]]:
    """Flags for the tables of one batch of blocks.

    ``volumes[c]`` is the device view of image channel ``c`` (raw voxels or a preprocessed slot
    buffer) addressed through ``blocks[i].src_off``; channels without a view cannot have blobs.
    ``tables[i]`` is block ``i``'s 11-column table with block-relative coordinates (or ``None``).
    ``means_only`` returns the ``(rows, n_channels)`` mean matrices (NaN where not computed) instead.
    """
    L = nat.lib()
    live = [i for i, t in enumerate(tables) if t is not None and len(t)]
    out: List[Optional[np.ndarray]] = [
        None if t is None else (np.full((len(t), n_channels), np.nan) if means_only
                                else np.zeros((len(t), n_channels), np.uint8)) for t in tables]
    if not live:
        return out
    counts = np.array([len(tables[i]) if i in set(live) else 0 for i in range(len(tables))], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    n = int(offsets[-1])
    rows = np.zeros((n, 5), dtype=np.int32)
    for i in live:
        t = tables[i]
        a = offsets[i]
        rows[a:a + len(t), 0] = i
        rows[a:a + len(t), 1:4] = t[:, :3].astype(int)          # the reference's .astype(int)
        rows[a:a + len(t), 4] = t[:, 6].astype(int)
    d_rows = torch.from_numpy(rows.reshape(-1)).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_mean = torch.empty(n, dtype=torch.float64, device=dev)
    d_cnt = torch.empty(n, dtype=torch.int32, device=dev)
    means = np.full((n, n_channels), np.nan)
    stream = torch.cuda.current_stream().cuda_stream
    for c, vol in sorted(volumes.items()):
        nat.check(L.mmx_coloc_means(ctypes.byref(vol), d_blocks.data_ptr(), len(blocks), d_rows.data_ptr(),
                                    d_off.data_ptr(), n, d_mean.data_ptr(), d_cnt.data_ptr(), stream),
                  "mmx_coloc_means")
        means[:, c] = d_mean.cpu().numpy()
    for i in live:
        a, b = offsets[i], offsets[i + 1]
        out[i] = means[a:b] if means_only else _flags_from_means(tables[i], means[a:b], shapes[i], n_channels)
    return out


def colocalize_blobs(roi, blobs: Optional[np.ndarray], thresh=None) -> Optional[np.ndarray]:
    """``(len(blobs), n_channels)`` uint8 flags for one ROI (same call as the reference's)."""
    from . import blob_log as bl
    if blobs is None or roi is None or len(roi.shape) < 4:
        return None
    if thresh is not None and thresh != "min":
        raise NotImplementedError("only the default thresh='min' is built on this path")
    dvol = roi if isinstance(roi, bl.DeviceVolume) else bl.DeviceVolume(roi)
    shape3 = tuple(dvol.shape[:3])
    blocks, _ = bl._make_blocks(dvol, 0, [(0, 0, 0)], [shape3])
    d_blocks = bl._to_device_bytes(blocks, dvol.tensor.device)
    volumes = {c: dvol.view(c, False) for c in range(dvol.n_channels)}
    return colocalize_blocks_device(volumes, blocks, d_blocks, [shape3], [np.asarray(blobs)],
                                    dvol.n_channels, dvol.tensor.device)[0]
