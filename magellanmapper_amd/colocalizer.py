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

Match-based co-localisation (reference :20-337, :444-501): :class:`BlobMatch`, :func:`colocalize_blobs_match`
and :class:`StackColocalizer` pair the blobs of every two channels by optimal assignment on their distances
(:mod:`verifier`: device distance matrices + the native assignment solver), block by block over a re-split
stack, and keep the shortest match of every blob matched more than once.  The reference's database
insertion of the matches (``insert_matches``, SQLite) is outside this path's scope.
"""
from __future__ import annotations

import ctypes
import warnings
from enum import Enum
from typing import Dict, List, Optional, Sequence, Tuple

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


# ------------------------------------------------------------------------- match-based co-localisation
class BlobMatch:
    """Blob matches as a data frame (reference colocalizer.py:20-162): one row per match with the two blob rows
    and their (scaled) distance; same column names as the reference so that its consumers read it."""

    class Cols(Enum):
        MATCH_ID = "MatchID"
        ROI_ID = "RoiID"
        BLOB1_ID = "Blob1ID"
        BLOB1 = "Blob1"
        BLOB2_ID = "Blob2ID"
        BLOB2 = "Blob2"
        DIST = "Distance"

    def __init__(self, matches=None, match_id=None, roi_id=None, blob1_id=None, blob2_id=None, df=None):
        import pandas as pd
        self.df = None
        self.coords = None
        self.cmap = None
        if df is not None:
            self.df = df
            return
        if matches is None:
            return
        n = len(matches)
        ids = {BlobMatch.Cols.MATCH_ID: match_id, BlobMatch.Cols.ROI_ID: roi_id,
               BlobMatch.Cols.BLOB1_ID: blob1_id, BlobMatch.Cols.BLOB2_ID: blob2_id}
        data = {}
        for col in BlobMatch.Cols:
            if col in ids:
                data[col.value] = [None] * n if ids[col] is None else list(ids[col])
            else:
                k = {BlobMatch.Cols.BLOB1: 0, BlobMatch.Cols.BLOB2: 1, BlobMatch.Cols.DIST: 2}[col]
                data[col.value] = [m[k] for m in matches]
        self.df = pd.DataFrame(data)

    def __repr__(self):
        return "Empty blob matches" if self.df is None else repr(self.df)

    def get_blobs(self, n: int) -> Optional[np.ndarray]:
        col = BlobMatch.Cols.BLOB1 if n == 1 else BlobMatch.Cols.BLOB2
        if self.df is None or col.value not in self.df or len(self.df[col.value]) == 0:
            return None
        return np.vstack(self.df[col.value])

    def get_blobs_all(self) -> Optional[List[np.ndarray]]:
        out = []
        for n in (1, 2):
            blobs = self.get_blobs(n)
            if blobs is None:
                return None
            out.append(blobs)
        return out

    def update_blobs(self, fn, *args):
        if self.df is None:
            return
        for i, col in enumerate((BlobMatch.Cols.BLOB1, BlobMatch.Cols.BLOB2)):
            blobs = self.get_blobs(i + 1)
            if blobs is not None:
                self.df[col.value] = fn(blobs, *args).tolist()

    def get_mean_coords(self):
        blobs = self.get_blobs_all()
        if blobs is None:
            return None
        self.coords = np.mean([b[:, :3] for b in blobs], axis=0)
        return self.coords


def colocalize_blobs_match(blobs, offset: Sequence[int], size: Sequence[int], tol: Sequence[float],
                           inner_padding: Optional[Sequence[int]] = None,
                           channels: Optional[Sequence[int]] = None) -> Optional[Dict[Tuple[int, int], BlobMatch]]:
    """Pair the blobs of every two channels inside one ROI (``offset`` / ``size`` / ``tol`` in x, y, z) by
    optimal assignment -> ``{(channel, other channel): BlobMatch}``, ``None`` without blobs
    (reference colocalizer.py:444-501)."""
    from . import verifier
    if blobs is None:
        return None
    thresh, scaling, inner_pad, resize, blobs_roi = verifier.setup_match_blobs_roi(tol, blobs)
    if inner_padding is None:
        inner_padding = inner_pad
    matches_chls = {}
    blob_chls = np.unique(blobs.get_blobs_channel(blobs_roi)).astype(int)
    if channels is not None:
        blob_chls = [c for c in blob_chls if c in channels]
    for chl in blob_chls:
        blobs_chl = blobs.blobs_in_channel(blobs_roi, chl)
        for chl_other in blob_chls:
            if chl >= chl_other:          # each pair once
                continue
            blobs_chl_other = blobs.blobs_in_channel(blobs_roi, chl_other)
            matches = verifier.match_blobs_roi(blobs_chl_other, blobs_chl, offset, size, thresh, scaling,
                                               inner_padding, resize)[4]
            matches.update_blobs(blobs.set_blob_truth, -1)
            matches.update_blobs(blobs.set_blob_confirmed, -1)
            matches_chls[(int(chl), int(chl_other))] = matches
    return matches_chls


class StackColocalizer:
    """Match-based co-localisation of a whole stack, block by block (reference colocalizer.py:165-337).  The
    reference fans the blocks out to a process pool; here they run on a thread pool of ``config.cpus`` workers
    (the distance matrices come from the device, the assignment solver is native code that releases the GIL)."""
    blobs = None
    match_tol = None
    channels = None

    @classmethod
    def colocalize_block(cls, coord, offset, shape, blobs=None, tol=None, setup_cli: bool = False, channels=None):
        blobs = cls.blobs if blobs is None else blobs
        tol = cls.match_tol if tol is None else tol
        channels = cls.channels if channels is None else channels
        matches = colocalize_blobs_match(blobs, offset[::-1], shape[::-1], tol, channels=channels)
        return coord, matches

    @classmethod
    def colocalize_stack(cls, shape: Sequence[int], blobs, channels: Optional[Sequence[int]] = None
                         ) -> Dict[Tuple[int, int], BlobMatch]:
        """``{(channel, other channel): BlobMatch}`` for the stack of ``shape`` (z, y, x)."""
        import pandas as pd
        from concurrent.futures import ThreadPoolExecutor
        from . import chunking, config, stack_detect, verifier
        blocks = stack_detect.setup_blocks(config.roi_profile, shape)
        match_tol = np.multiply(blocks.overlap_base, config.roi_profile["verify_tol_factor"])
        # blocks with the inner padding of the matcher on top of the raw overlap
        inner_pad = np.add(verifier.setup_match_blobs_roi(match_tol)[2], blocks.overlap_base)
        sub_roi_slices, sub_rois_offsets = chunking.stack_splitter(shape, blocks.max_pixels, inner_pad[::-1])
        jobs = []
        for coord in np.ndindex(*sub_roi_slices.shape):
            slices = sub_roi_slices[coord]
            jobs.append((coord, sub_rois_offsets[coord], [s.stop - s.start for s in slices]))
        workers = max(1, int(config.cpus or 1))
        with ThreadPoolExecutor(max_workers=workers) as pool:
            results = list(pool.map(lambda j: cls.colocalize_block(j[0], j[1], j[2], blobs, match_tol, False,
                                                                   channels), jobs))
        matches_all: Dict[Tuple[int, int], list] = {}
        for _, matches in results:                       # block order, as the reference collects them
            for key, val in matches.items():
                matches_all.setdefault(key, []).append(val.df)
        # blobs matched in several blocks (or to several partners) keep their shortest match, first of equals
        for key in matches_all:
            matches = pd.concat(matches_all[key])
            if matches.size > 0:
                for blobi in (BlobMatch.Cols.BLOB1, BlobMatch.Cols.BLOB2):
                    coords = np.vstack(matches[blobi.value])[:, :3]
                    _, first, inv, counts = np.unique(coords, axis=0, return_index=True, return_inverse=True,
                                                      return_counts=True)
                    inv = np.asarray(inv).reshape(-1)
                    if np.sum(counts > 1) > 0:
                        dist = matches[BlobMatch.Cols.DIST.value].to_numpy()
                        keep = list(first[counts == 1])
                        for i in np.nonzero(counts > 1)[0]:
                            rows = np.nonzero(inv == i)[0]
                            keep.append(rows[dist[rows] == np.amin(dist[rows])][0])
                        matches = matches.iloc[np.asarray(keep, dtype=int)]
            matches_all[key] = BlobMatch(df=matches)
        return matches_all
