"""Blob matching between two sets (the parts of ``magmap.cv.verifier`` that match-based co-localisation
needs).

Public functions keep the reference's names, arguments and return values (they are the interface
``colocalizer.colocalize_blobs_match`` and the reference's GUI call):

* :func:`find_closest_blobs_cdist` -- reference magmap/cv/verifier.py:47-119.  Its two third-party calls are
  replaced: ``scipy.spatial.distance.cdist`` by ``mmx_cdist_f64`` (HIP, bit-equal float64) and
  ``scipy.optimize.linear_sum_assignment`` by ``mmx_host_lsap`` (native solver that returns SciPy's optimum
  also where distances tie).  Neither SciPy routine is imported here.
* :func:`setup_match_blobs_roi` -- :122-160.
* :func:`match_blobs_roi` -- :164-289.

How the ROI match is built here.  The reference carries its bookkeeping in columns 4 ("confirmed") and 5
("truth") of copies of the tables; this file keeps the tables untouched and works on row numbers:

1. ``_Box`` gives the member rows of the ROI and of its core (the ROI shrunk by the inner padding).
2. Round one assigns the detections of the core to every base blob of the ROI; a :class:`_Round` records the
   pairs as row numbers into the ROI's member lists.
3. A base blob is *open* after round one when its truth flag reads 0: core members start at 0, the others keep
   what the caller's table says, round one's partners are closed.  Round two assigns the detections of the rim
   to the open base blobs.
4. The four returned tables and the match rows are cut from the originals by those row numbers, and only then
   do the two flag columns get the values the reference leaves in them.

Truth-set verification against a database (``verify_rois`` and friends) is outside this path's scope.
"""
from __future__ import annotations

import threading
from typing import NamedTuple, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from . import config

#: columns of a blob row that carry the match flags (``detector.Blobs.Cols.CONFIRMED`` / ``TRUTH``)
_COL_CONFIRMED, _COL_TRUTH = 4, 5
_device_lock = threading.Lock()      # cdist launches of concurrent block workers take turns on the stream


def _cdist(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Euclidean distance matrix in float64 on the device, bit-equal to ``scipy.spatial.distance.cdist(a, b)``
    (rows of up to 64 values: scaled z, y, x on this path, whole blob rows when a caller passes no scaling)."""
    n, m = len(a), len(b)
    if n == 0 or m == 0:
        return np.zeros((n, m))
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if a.shape[1] != b.shape[1]:
        raise ValueError("XA and XB must have the same number of columns")
    import torch
    from . import blob_log as bl
    dev = bl._require_gpu()
    out = np.empty((n, m))
    with _device_lock:
        d_b = torch.from_numpy(b).to(dev)
        for lo in range(0, n, 32768):                      # (grid.y limit of one launch)
            hi = min(n, lo + 32768)
            d_a = torch.from_numpy(a[lo:hi]).to(dev)
            d_out = torch.empty((hi - lo, m), dtype=torch.float64, device=dev)
            nat.check(nat.lib().mmx_cdist_f64(d_a.data_ptr(), hi - lo, d_b.data_ptr(), m, a.shape[1],
                                              d_out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                      "mmx_cdist_f64")
            out[lo:hi] = d_out.cpu().numpy()
    return out


def linear_sum_assignment(cost: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Optimal assignment of a dense cost matrix: ``min(n, m)`` (row, column) pairs in ascending row order
    (``mmx_host_lsap``; the GIL is released while it runs)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    n, m = cost.shape
    k = min(n, m)
    rows = np.empty(k, dtype=np.int64)
    cols = np.empty(k, dtype=np.int64)
    nat.check(nat.lib().mmx_host_lsap(cost.ctypes.data, n, m, rows.ctypes.data, cols.ctypes.data), "mmx_host_lsap")
    return rows, cols


def find_closest_blobs_cdist(blobs: np.ndarray, blobs_master: np.ndarray, thresh: Optional[float] = None,
                             scaling: Optional[Sequence[float]] = None
                             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One-to-one pairing of ``blobs`` with ``blobs_master`` that minimises the summed distance ->
    ``(row numbers, column numbers, distances)``; with ``thresh`` only the pairs closer than it.  ``scaling``
    multiplies the leading coordinates (and drops the other columns) before distances are taken."""
    pts, pts_master = blobs, blobs_master
    if scaling is not None:
        dims = len(scaling)
        pts = np.multiply(blobs[:, :dims], scaling)
        pts_master = np.multiply(blobs_master[:, :dims], scaling)
    cost = _cdist(pts, pts_master)
    rows, cols = linear_sum_assignment(cost)
    lengths = cost[rows, cols]
    if thresh is None:
        return rows, cols, lengths
    near = np.flatnonzero(lengths < thresh)
    return rows[near], cols[near], lengths[near]


def setup_match_blobs_roi(tol: Sequence[float], blobs: Optional["detector.Blobs"] = None):
    """Matching parameters from per-axis tolerances -> ``(thresh, scaling, inner_padding, resize,
    blobs_roi)``: the distance threshold is the largest tolerance, ``scaling`` stretches every axis so that its
    tolerance becomes that threshold, the inner padding is the tolerance rounded down, in reversed axis
    order.  ``resize`` is the first profile's ``resize_blobs``; when set, ``blobs_roi`` is the resized table."""
    tol = np.asarray(tol, dtype=float)
    thresh = tol.max()
    resize = config.get_roi_profile(0)["resize_blobs"]
    table = blobs.blobs if blobs is not None else None
    if table is not None and resize:
        table = blobs.multiply_blob_rel_coords(table, resize)
    return thresh, thresh / tol, np.floor(tol[::-1]), resize, table


class _Box(NamedTuple):
    """Half-open box in z, y, x."""
    lo: np.ndarray
    hi: np.ndarray

    @classmethod
    def from_xyz(cls, offset, size) -> "_Box":
        lo = np.asarray(offset, dtype=float)[::-1]
        return cls(lo, lo + np.asarray(size, dtype=float)[::-1])

    def members(self, table: np.ndarray) -> np.ndarray:
        """Row numbers of ``table`` whose z, y, x lie inside the box, ascending."""
        zyx = table[:, :3]
        return np.flatnonzero(np.all((zyx >= self.lo) & (zyx < self.hi), axis=1))


class _Round(NamedTuple):
    """One assignment round: pair ``k`` joins detection row ``det[k]`` and base row ``base[k]`` (row numbers
    into the two tables the round was run on) at scaled distance ``dist[k]``."""
    det: np.ndarray
    base: np.ndarray
    dist: np.ndarray

    def by_base_position(self, base_rows: np.ndarray) -> "_Round":
        """The pairs re-ordered by the base blob's z, then y, then x (how the reference lists matches)."""
        order = np.lexsort((base_rows[self.base, 2], base_rows[self.base, 1], base_rows[self.base, 0]))
        return _Round(self.det[order], self.base[order], self.dist[order])


def _flagged(rows: np.ndarray, col: int, value) -> np.ndarray:
    """A copy of ``rows`` with one flag column set."""
    out = np.array(rows, copy=True)
    out[:, col] = value
    return out


def match_blobs_roi(blobs: np.ndarray, blobs_base: np.ndarray, offset, size, thresh: float, scaling,
                    inner_padding, resize=None):
    """Match ``blobs`` (detections) against ``blobs_base`` inside the ROI ``offset`` / ``size`` (x, y, z) ->
    ``(blobs_inner_plus, blobs_truth_inner_plus, offset_inner, size_inner, matches)``.

    ``blobs_inner_plus``: the core's detections (confirmed 1 where paired, else 0) followed by the rim's paired
    detections; ``blobs_truth_inner_plus``: base blobs flagged as found followed by the base blobs that round
    two worked on; ``matches``: a :class:`colocalizer.BlobMatch` of ``(base row, detection row, distance)``,
    round one's pairs then round two's, each sorted by the base blob's position.
    """
    from . import colocalizer
    if resize is not None:
        raise NotImplementedError("resize_blobs is a visualisation setting outside this path's scope")
    # the core: the ROI minus a symmetric margin that always leaves something in the middle
    half = np.ceil(np.divide(size, 2) - 1)
    margin = np.clip(inner_padding, 0, np.clip(half, 0, None))
    offset_inner = np.add(offset, margin)
    size_inner = np.subtract(size, margin * 2)
    roi, core = _Box.from_xyz(offset, size), _Box.from_xyz(offset_inner, size_inner)

    det = blobs[roi.members(blobs)]                # detections / base blobs of the ROI, table order
    base = blobs_base[roi.members(blobs_base)]
    det_core = core.members(det)
    det_rim = np.setdiff1d(np.arange(len(det)), det_core, assume_unique=True)
    base_core = core.members(base)

    # round one: the core's detections against every base blob of the ROI
    one = _Round(*find_closest_blobs_cdist(det[det_core], base, thresh, scaling))
    truth = base[:, _COL_TRUTH].copy()             # the truth flags as the reference would leave them
    truth[base_core] = 0
    truth[one.base] = 1
    # round two: base blobs still open against the rim's detections
    open_rows = np.flatnonzero(truth == 0)
    two = _Round(*find_closest_blobs_cdist(det[det_rim], base[open_rows], thresh, scaling))

    confirmed_core = np.zeros(len(det_core))
    confirmed_core[one.det] = 1
    core_rows = _flagged(det[det_core], _COL_CONFIRMED, confirmed_core)
    rim_paired = _flagged(det[det_rim[two.det]], _COL_CONFIRMED, 1)
    base_now = _flagged(base, _COL_TRUTH, truth)
    truth_open = np.zeros(len(open_rows))
    truth_open[two.base] = 1
    open_now = _flagged(base[open_rows], _COL_TRUTH, truth_open)

    blobs_inner_plus = np.concatenate((core_rows, rim_paired))
    blobs_truth_inner_plus = np.concatenate((base_now[truth == 1], open_now))
    one, two = one.by_base_position(base_now), two.by_base_position(open_now)
    two_rows = _flagged(det[det_rim[two.det]], _COL_CONFIRMED, 1)      # (rim_paired in the sorted order)
    matches = colocalizer.BlobMatch.from_arrays(
        np.concatenate((base_now[one.base], open_now[two.base])),
        np.concatenate((core_rows[one.det], two_rows)),
        np.concatenate((one.dist, two.dist)))
    return blobs_inner_plus, blobs_truth_inner_plus, offset_inner, size_inner, matches
