"""Blob matching between two sets (mirror of the parts of ``magmap.cv.verifier`` the match-based
co-localisation uses).

* :func:`find_closest_blobs_cdist` -- reference magmap/cv/verifier.py:47-119: full distance matrix, optimal
  assignment, threshold.  The two third-party calls there (``scipy.spatial.distance.cdist``,
  ``scipy.optimize.linear_sum_assignment``) are replaced by ``mmx_cdist_f64`` (HIP, bit-equal float64) and
  ``mmx_host_lsap`` (native shortest-augmenting-path solver that returns SciPy's optimum also where distances
  tie); neither SciPy routine is imported here.
* :func:`setup_match_blobs_roi` -- :122-160; :func:`match_blobs_roi` -- :164-289: inner blobs first against all
  base blobs of the ROI, base blobs still unmatched then against the blobs of the outer rim.
Truth-set verification against a database (``verify_rois`` and friends) is outside this path's scope.
"""
from __future__ import annotations

import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from . import config, detector

_device_lock = threading.Lock()      # cdist launches of concurrent block workers take turns on the stream


def _cdist(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``scipy.spatial.distance.cdist(a, b)`` (Euclidean, float64) on the device."""
    import torch
    from . import blob_log as bl
    n, m = len(a), len(b)
    if n == 0 or m == 0:
        return np.zeros((n, m))
    dev = bl._require_gpu()
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
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
    """Closest ``blobs`` -> ``blobs_master`` matches by optimal assignment; pairs at or beyond ``thresh`` are
    dropped -> ``(rows, columns, distances)``."""
    scaled, scaled_master = blobs, blobs_master
    if scaling is not None:
        n = len(scaling)
        scaled = np.multiply(blobs[:, :n], scaling)
        scaled_master = np.multiply(blobs_master[:, :n], scaling)
    dists = _cdist(scaled, scaled_master)
    rowis, colis = linear_sum_assignment(dists)
    closest = dists[rowis, colis]
    if thresh is not None:
        inside = closest < thresh
        rowis, colis, closest = rowis[inside], colis[inside], closest[inside]
    return rowis, colis, closest


def setup_match_blobs_roi(tol: Sequence[float], blobs: Optional["detector.Blobs"] = None):
    """``(thresh, scaling, inner_padding, resize, blobs_roi)``: the largest tolerance as the distance
    threshold, coordinates scaled so that every axis' tolerance becomes that threshold, ``floor(tol)`` in
    x, y, z as the inner padding; blobs resized by the first profile's ``resize_blobs`` if set."""
    tol = np.asarray(tol, dtype=float)
    thresh = np.amax(tol)
    scaling = thresh / tol
    inner_padding = np.floor(tol[::-1])
    resize = config.get_roi_profile(0)["resize_blobs"]
    blobs_roi = None if blobs is None else blobs.blobs
    if resize and blobs_roi is not None:
        blobs_roi = blobs.multiply_blob_rel_coords(blobs_roi, resize)
    return thresh, scaling, inner_padding, resize, blobs_roi


def _match_blobs(blobs, blobs_master, close, close_master, dists) -> List[tuple]:
    """``(master, blob, distance)`` triples sorted by the master's z, y, x (reference verifier.py:23-44)."""
    found_master, order = detector.sort_blobs(blobs_master[close_master])
    found = blobs[close][order]
    return [(fm, f, d) for f, fm, d in zip(found, found_master, np.asarray(dists)[order])]


def match_blobs_roi(blobs: np.ndarray, blobs_base: np.ndarray, offset, size, thresh: float, scaling,
                    inner_padding, resize=None):
    """Match ``blobs`` against ``blobs_base`` inside the ROI ``offset`` / ``size`` (x, y, z) ->
    ``(blobs_inner_plus, blobs_truth_inner_plus, offset_inner, size_inner, matches)`` with ``matches`` a
    :class:`colocalizer.BlobMatch`."""
    from . import colocalizer
    if resize is not None:
        raise NotImplementedError("resize_blobs is a visualisation setting outside this path's scope")
    inner_padding_max = np.clip(np.ceil(np.divide(size, 2) - 1), 0, None)
    inner_padding = np.clip(inner_padding, 0, inner_padding_max)
    size_inner = np.subtract(size, inner_padding * 2)
    offset_inner = np.add(offset, inner_padding)
    blobs_roi, _ = detector.get_blobs_in_roi(blobs, offset, size)
    blobs_inner, blobs_inner_mask = detector.get_blobs_in_roi(blobs_roi, offset_inner, size_inner)
    blobs_base_roi, _ = detector.get_blobs_in_roi(blobs_base, offset, size)
    _, blobs_base_inner_mask = detector.get_blobs_in_roi(blobs_base_roi, offset_inner, size_inner)

    # inner blobs against every base blob of the ROI, closest first
    found, found_base, dists = find_closest_blobs_cdist(blobs_inner, blobs_base_roi, thresh, scaling)
    blobs_inner[:, 4] = 0
    blobs_inner[found, 4] = 1
    blobs_base_roi[blobs_base_inner_mask, 5] = 0
    blobs_base_roi[found_base, 5] = 1
    # base blobs missed so far get a second chance against the blobs of the outer rim
    blobs_base_inner_missed = blobs_base_roi[blobs_base_roi[:, 5] == 0]
    blobs_outer = blobs_roi[np.invert(blobs_inner_mask)]
    found_out, found_base_out, dists_out = find_closest_blobs_cdist(blobs_outer, blobs_base_inner_missed, thresh,
                                                                    scaling)
    blobs_base_inner_missed[found_base_out, 5] = 1
    blobs_truth_inner_plus = np.concatenate((blobs_base_roi[blobs_base_roi[:, 5] == 1], blobs_base_inner_missed))
    blobs_outer[found_out, 4] = 1
    blobs_inner_plus = np.concatenate((blobs_inner, blobs_outer[found_out]))
    matches_inner = _match_blobs(blobs_inner, blobs_base_roi, found, found_base, dists)
    matches_outer = _match_blobs(blobs_outer, blobs_base_inner_missed, found_out, found_base_out, dists_out)
    matches = colocalizer.BlobMatch([*matches_inner, *matches_outer])
    return blobs_inner_plus, blobs_truth_inner_plus, offset_inner, size_inner, matches
