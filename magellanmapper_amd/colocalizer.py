"""Intensity co-localisation of blobs across channels (mirror of ``magmap.cv.colocalizer``'s
``colocalize_blobs``, reference magmap/cv/colocalizer.py:340-441).

A blob of channel A "co-localises" with channel B when the mean intensity of B over the voxels
the blob owns (a ``ball(2)`` around its centre; voxels contested by several blobs of the same
channel go to the higher table row) reaches B's threshold -- the smallest such mean among B's
own blobs.  The per-blob means come from the device (``mmx_coloc_means``, bit-equal float64 in
NumPy's summation order); thresholds and flags are a handful of NumPy reductions on the
``(n_blobs, n_channels)`` matrix.

The block path (``StackDetector.detect_sub_roi``, stack_detect.py:159-162) uses the default ``thresh="min"``;
a number asks for that percentile of a channel's intensities over all voxels its own blobs own instead
(:403-409): the device hands those voxels out (``mmx_coloc_voxels``), ``np.percentile`` takes it from there.

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


def _flags_from_means(table: np.ndarray, means: np.ndarray, shape3, n_channels: int,
                      thresholds: Optional[Dict[int, float]] = None) -> np.ndarray:
    """Thresholds + flags of one block: ``means[b, c]`` = mean of channel ``c`` over blob ``b``'s
    voxels (NaN when it owns none).  Rows outside the ROI get zeros (colocalizer.py:375-378, 434-436).
    ``thresholds``: per-channel thresholds to use instead of the smallest mean of the channel's own blobs."""
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
        if thresholds is not None:
            thr = thresholds[int(other)]
        else:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                thr = np.amin(means[own, other])      # NaN (a blob that owns nothing) poisons it
        hit = in_roi & np.isin(chl, present) & (means[:, other] >= thr)
        colocs[hit, other] = 1
    return colocs


def colocalize_blocks_device(volumes: Dict[int, nat.Volume], blocks, d_blocks,
                             shapes, tables: List[Optional[np.ndarray]], n_channels: int,
                             dev, means_only: bool = False, percentile=None) -> List[Optional[np.ndarray]]:
    """Flags for the tables of one batch of blocks.

    ``volumes[c]`` is the device view of image channel ``c`` (raw voxels or a preprocessed slot
    buffer) addressed through ``blocks[i].src_off``; channels without a view cannot have blobs.
    ``blocks`` / ``d_blocks``: one block table for every channel, or dicts of them per channel (preprocessed
    channels live in slot buffers of their own).  The kernels of all channels are queued before the host waits once.
    ``tables[i]`` is block ``i``'s 11-column table with block-relative coordinates (or ``None``).
    ``means_only`` returns the ``(rows, n_channels)`` mean matrices (NaN where not computed) instead.
    ``percentile``: a channel's threshold is this percentile of its intensities over every voxel owned by one of
    its own in-ROI blobs of the block (reference :403-409) instead of the smallest per-blob mean.
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
    from . import blob_log as _bl
    d_rows = _bl.to_device(rows.reshape(-1), dev)
    d_off = _bl.to_device(offsets, dev)
    order = sorted(volumes)
    d_mean = torch.empty((len(order), n), dtype=torch.float64, device=dev)
    d_cnt = torch.empty(n, dtype=torch.int32, device=dev)
    means = np.full((n, n_channels), np.nan)
    stream = torch.cuda.current_stream().cuda_stream
    owned: Dict[int, Tuple[np.ndarray, np.ndarray]] = {}
    d_vox = torch.empty((n, nat.MMX_COLOC_BALL), dtype=torch.float64, device=dev) if percentile is not None else None
    for k, c in enumerate(order):
        vol = volumes[c]
        blk = blocks[c] if isinstance(blocks, dict) else blocks
        d_blk = d_blocks[c] if isinstance(d_blocks, dict) else d_blocks
        if d_vox is None:
            nat.check(L.mmx_coloc_means(ctypes.byref(vol), d_blk.data_ptr(), len(blk), d_rows.data_ptr(),
                                        d_off.data_ptr(), n, d_mean[k].data_ptr(), d_cnt.data_ptr(), stream),
                      "mmx_coloc_means")
        else:
            nat.check(L.mmx_coloc_voxels(ctypes.byref(vol), d_blk.data_ptr(), len(blk), d_rows.data_ptr(),
                                         d_off.data_ptr(), n, d_mean[k].data_ptr(), d_cnt.data_ptr(),
                                         d_vox.data_ptr(), stream), "mmx_coloc_voxels")
            owned[c] = (d_vox.cpu().numpy(), d_cnt.cpu().numpy())
    h_mean = d_mean.cpu().numpy()           # (one wait for every channel's kernel)
    for k, c in enumerate(order):
        means[:, c] = h_mean[k]
    for i in live:
        a, b = offsets[i], offsets[i + 1]
        if means_only:
            out[i] = means[a:b]
            continue
        thresholds = None
        if percentile is not None:
            t, shp = tables[i], shapes[i]
            in_roi = np.all([t[:, k] >= 0 for k in range(3)] + [t[:, k] < shp[k] for k in range(3)], axis=0)
            thresholds = {}
            for c, (vox, cnt) in owned.items():
                mine = np.flatnonzero(in_roi & (t[:, 6].astype(int) == c)) + a
                if len(mine):        # (the channel's last blob always owns its own centre: never empty)
                    vals = np.concatenate([vox[r, :cnt[r]] for r in mine])
                    thresholds[c] = np.percentile(vals, percentile)
        out[i] = _flags_from_means(tables[i], means[a:b], shapes[i], n_channels, thresholds)
    return out


def colocalize_blobs(roi, blobs: Optional[np.ndarray], thresh=None) -> Optional[np.ndarray]:
    """``(len(blobs), n_channels)`` uint8 flags for one ROI (same call as the reference's)."""
    from . import blob_log as bl
    if blobs is None or roi is None or len(roi.shape) < 4:
        return None
    percentile = None if thresh is None or (isinstance(thresh, str) and thresh == "min") else float(thresh)
    dvol = roi if isinstance(roi, bl.DeviceVolume) else bl.DeviceVolume(roi)
    dvol.wait_all()                 # (a volume still on its way up: the whole ROI is read here)
    shape3 = tuple(dvol.shape[:3])
    blocks, _ = bl._make_blocks(dvol, 0, [(0, 0, 0)], [shape3])
    d_blocks = bl._to_device_bytes(blocks, dvol.tensor.device)
    volumes = {c: dvol.view(c, False) for c in range(dvol.n_channels)}
    return colocalize_blocks_device(volumes, blocks, d_blocks, [shape3], [np.asarray(blobs)],
                                    dvol.n_channels, dvol.tensor.device, percentile=percentile)[0]


# ------------------------------------------------------------------------- match-based co-localisation
class _PairTable:
    """The rows of a :class:`BlobMatch` as arrays: ``first`` / ``second`` hold one blob row per match (the base
    channel's blob and its partner), ``dist`` their distance, ``ids`` the optional id columns."""
    __slots__ = ("first", "second", "dist", "ids")

    ID_COLS = ("MatchID", "RoiID", "Blob1ID", "Blob2ID")

    def __init__(self, first: np.ndarray, second: np.ndarray, dist: np.ndarray, ids: Optional[dict] = None):
        self.dist = np.asarray(dist, dtype=np.float64).reshape(-1)
        self.first = self._rows(first, len(self.dist))
        self.second = self._rows(second, len(self.dist))
        self.ids = {k: (None if v is None else list(v)) for k, v in (ids or {}).items()}

    @staticmethod
    def _rows(rows, n: int) -> np.ndarray:
        rows = np.asarray(rows, dtype=np.float64)
        if n == 0:
            return np.zeros((0, rows.shape[1] if rows.ndim == 2 else 0))
        return rows.reshape(n, -1)

    def __len__(self) -> int:
        return len(self.dist)

    def take(self, rows: np.ndarray) -> "_PairTable":
        rows = np.asarray(rows, dtype=np.int64)
        ids = {k: (None if v is None else [v[i] for i in rows]) for k, v in self.ids.items()}
        return _PairTable(self.first[rows], self.second[rows], self.dist[rows], ids)

    @classmethod
    def joined(cls, parts: Sequence["_PairTable"]) -> "_PairTable":
        parts = [p for p in parts if p is not None and len(p)]
        if not parts:
            return cls(np.zeros((0, 0)), np.zeros((0, 0)), np.zeros(0))
        ids = {}
        for name in cls.ID_COLS:
            if any(p.ids.get(name) is not None for p in parts):
                ids[name] = [v for p in parts for v in (p.ids.get(name) or [None] * len(p))]
        return cls(np.concatenate([p.first for p in parts]), np.concatenate([p.second for p in parts]),
                   np.concatenate([p.dist for p in parts]), ids)

    @classmethod
    def from_frame(cls, frame) -> "_PairTable":
        n = len(frame)
        if n == 0 or "Blob1" not in frame or "Blob2" not in frame:
            return cls(np.zeros((0, 0)), np.zeros((0, 0)), np.zeros(0))
        ids = {name: frame[name].tolist() for name in cls.ID_COLS if name in frame}
        dist = frame["Distance"].to_numpy(dtype=float) if "Distance" in frame else np.full(n, np.nan)
        return cls(np.vstack(frame["Blob1"].tolist()), np.vstack(frame["Blob2"].tolist()), dist, ids)

    def to_frame(self):
        """The data frame the reference's consumers read: one row per match, the seven ``BlobMatch.Cols`` in
        their order, blob rows as arrays; no columns at all without matches (what a frame made from an empty
        dictionary looks like)."""
        import pandas as pd
        n = len(self)
        if n == 0:
            return pd.DataFrame()
        none = [None] * n
        ids = {name: (self.ids.get(name) or none) for name in self.ID_COLS}
        return pd.DataFrame({"MatchID": ids["MatchID"], "RoiID": ids["RoiID"], "Blob1ID": ids["Blob1ID"],
                             "Blob1": list(self.first), "Blob2ID": ids["Blob2ID"], "Blob2": list(self.second),
                             "Distance": self.dist})


class BlobMatch:
    """Blob matches of one channel pair (the interface of the reference's class, colocalizer.py:20-162:
    ``Cols``, ``df``, ``coords``, ``cmap``, ``get_blobs``, ``get_blobs_all``, ``update_blobs``,
    ``get_mean_coords``).  The matches live in a :class:`_PairTable` of arrays; ``df`` renders them as the data
    frame the reference keeps (same column names and order), and assigning a frame to ``df`` reads it back in."""

    class Cols(Enum):
        MATCH_ID = "MatchID"
        ROI_ID = "RoiID"
        BLOB1_ID = "Blob1ID"
        BLOB1 = "Blob1"
        BLOB2_ID = "Blob2ID"
        BLOB2 = "Blob2"
        DIST = "Distance"

    def __init__(self, matches=None, match_id=None, roi_id=None, blob1_id=None, blob2_id=None, df=None):
        self.coords: Optional[np.ndarray] = None
        self.cmap: Optional[np.ndarray] = None
        self._pairs: Optional[_PairTable] = None
        self._frame = None
        if df is not None:                      # a frame wins over every other argument
            self.df = df
        elif matches is not None:
            triples = list(matches)
            ids = dict(MatchID=match_id, RoiID=roi_id, Blob1ID=blob1_id, Blob2ID=blob2_id)
            self._pairs = _PairTable([t[0] for t in triples], [t[1] for t in triples],
                                     [t[2] for t in triples], ids)

    @classmethod
    def from_arrays(cls, blob1: np.ndarray, blob2: np.ndarray, dist: np.ndarray) -> "BlobMatch":
        """Matches given as two equally long blob tables and their distances."""
        out = cls()
        out._pairs = _PairTable(blob1, blob2, dist)
        return out

    @property
    def pairs(self) -> Optional[_PairTable]:
        return self._pairs

    @property
    def df(self):
        if self._frame is None and self._pairs is not None:
            self._frame = self._pairs.to_frame()
        return self._frame

    @df.setter
    def df(self, frame):
        self._frame = frame
        self._pairs = None if frame is None else _PairTable.from_frame(frame)

    def __len__(self) -> int:
        return 0 if self._pairs is None else len(self._pairs)

    def __repr__(self):
        return "Empty blob matches" if self._pairs is None else repr(self.df)

    def get_blobs(self, n: int) -> Optional[np.ndarray]:
        """The blob rows of side ``n`` (1: the base channel's blobs, anything else: their partners); ``None``
        without matches."""
        if not len(self):
            return None
        return self._pairs.first if n == 1 else self._pairs.second

    def get_blobs_all(self) -> Optional[List[np.ndarray]]:
        return [self._pairs.first, self._pairs.second] if len(self) else None

    def update_blobs(self, fn, *args):
        """Replace both sides' blob rows by ``fn(rows, *args)``."""
        if not len(self):
            return
        self._pairs.first = np.asarray(fn(self._pairs.first, *args), dtype=np.float64)
        self._pairs.second = np.asarray(fn(self._pairs.second, *args), dtype=np.float64)
        self._frame = None

    def get_mean_coords(self):
        """``(n, 3)`` midpoints of the matched pairs (also kept in ``coords``); ``None`` without matches."""
        if not len(self):
            return None
        self.coords = (self._pairs.first[:, :3] + self._pairs.second[:, :3]) / 2
        return self.coords


def colocalize_blobs_match(blobs, offset: Sequence[int], size: Sequence[int], tol: Sequence[float],
                           inner_padding: Optional[Sequence[int]] = None,
                           channels: Optional[Sequence[int]] = None) -> Optional[Dict[Tuple[int, int], BlobMatch]]:
    """Pair the blobs of every two channels inside one ROI (``offset`` / ``size`` / ``tol`` in x, y, z) by
    optimal assignment -> ``{(channel, higher channel): BlobMatch}``, ``None`` without blobs
    (reference colocalizer.py:444-501).  The lower channel is the base side of each pair; the matched rows come
    back with their confirmed / truth flags reset to -1."""
    from itertools import combinations
    from . import verifier
    if blobs is None:
        return None
    thresh, scaling, default_padding, resize, table = verifier.setup_match_blobs_roi(tol, blobs)
    padding = default_padding if inner_padding is None else inner_padding
    present = [int(c) for c in np.unique(blobs.get_blobs_channel(table))]
    if channels is not None:
        present = [c for c in present if c in channels]
    per_channel = {c: blobs.blobs_in_channel(table, c) for c in present}
    out: Dict[Tuple[int, int], BlobMatch] = {}
    for low, high in combinations(present, 2):           # ascending channels: (0, 1), (0, 2), (1, 2) ...
        found = verifier.match_blobs_roi(per_channel[high], per_channel[low], offset, size, thresh, scaling,
                                         padding, resize)[4]
        found.update_blobs(blobs.set_blob_truth, -1)
        found.update_blobs(blobs.set_blob_confirmed, -1)
        out[(low, high)] = found
    return out


def _keep_shortest(pairs: _PairTable, side: str) -> _PairTable:
    """Blobs of one ``side`` (``"first"`` / ``"second"``) that appear in several matches -- found again in a
    neighbouring block, or paired with different partners there -- keep the shortest one, the earliest of equally
    short ones.  Row order as the reference leaves it (colocalizer.py:301-331): untouched when nothing repeats,
    otherwise the blobs matched once in ascending z, y, x, then the repeated ones in ascending z, y, x."""
    zyx = getattr(pairs, side)[:, :3]
    n = len(zyx)
    by_pos = np.lexsort((np.arange(n), zyx[:, 2], zyx[:, 1], zyx[:, 0]))          # position, then table order
    srt = zyx[by_pos]
    starts = np.flatnonzero(np.concatenate(([True], np.any(srt[1:] != srt[:-1], axis=1))))
    sizes = np.diff(np.append(starts, n))
    if sizes.max(initial=1) <= 1:
        return pairs
    group = np.repeat(np.arange(len(starts)), sizes)            # group of every sorted row
    # inside a group: shortest distance first, table order among equals
    by_len = np.lexsort((by_pos, pairs.dist[by_pos], group))
    best = by_pos[by_len[starts]]                                 # (groups keep their start after the re-sort)
    once = sizes == 1
    return pairs.take(np.concatenate((best[once], best[~once])))


class StackColocalizer:
    """Match-based co-localisation of a whole stack, block by block (reference colocalizer.py:165-337).  The
    reference fans the blocks out to a process pool; here they run on a thread pool of ``config.cpus`` workers
    (the distance matrices come from the device, the assignment solver is native code that releases the GIL)."""
    blobs = None
    match_tol = None
    channels = None

    @classmethod
    def colocalize_block(cls, coord, offset, shape, blobs=None, tol=None, setup_cli: bool = False, channels=None):
        """One block (``offset`` / ``shape`` in z, y, x) -> ``(coord, {channel pair: BlobMatch})``; arguments left
        ``None`` fall back to the class attributes, as in the reference's worker."""
        blobs = cls.blobs if blobs is None else blobs
        tol = cls.match_tol if tol is None else tol
        channels = cls.channels if channels is None else channels
        return coord, colocalize_blobs_match(blobs, offset[::-1], shape[::-1], tol, channels=channels)

    @classmethod
    def colocalize_stack(cls, shape: Sequence[int], blobs, channels: Optional[Sequence[int]] = None
                         ) -> Dict[Tuple[int, int], BlobMatch]:
        """``{(channel, higher channel): BlobMatch}`` for the stack of ``shape`` (z, y, x)."""
        from concurrent.futures import ThreadPoolExecutor
        from . import chunking, config, stack_detect, verifier
        geometry = stack_detect.setup_blocks(config.roi_profile, shape)
        tol = np.multiply(geometry.overlap_base, config.roi_profile["verify_tol_factor"])
        # the stack is cut again: blocks overlap by the matcher's inner padding on top of the detection overlap
        reach = np.add(verifier.setup_match_blobs_roi(tol)[2], geometry.overlap_base)
        cuts, corners = chunking.stack_splitter(shape, geometry.max_pixels, reach[::-1])

        def work(coord):
            extent = [s.stop - s.start for s in cuts[coord]]
            return cls.colocalize_block(coord, corners[coord], extent, blobs, tol, False, channels)[1]

        with ThreadPoolExecutor(max_workers=max(1, int(config.cpus or 1))) as pool:
            per_block = list(pool.map(work, list(np.ndindex(*cuts.shape))))       # grid order, z slowest
        gathered: Dict[Tuple[int, int], list] = {}
        for found in per_block:
            for pair, match in (found or {}).items():
                gathered.setdefault(pair, []).append(match.pairs)
        out = {}
        for pair, parts in gathered.items():
            table = _PairTable.joined(parts)
            for side in ("first", "second"):
                if len(table):
                    table = _keep_shortest(table, side)
            match = BlobMatch()
            match._pairs = table
            out[pair] = match
        return out
