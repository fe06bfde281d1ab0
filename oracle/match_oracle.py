"""Match-based co-localisation restated (TEST INFRASTRUCTURE: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s CPU baseline may import this package; the product path never does).

What the reference does (the part that was still missing):

* ``verifier.find_closest_blobs_cdist`` (magmap/cv/verifier.py:47-119): scale the coordinates, full
  Euclidean distance matrix (``scipy.spatial.distance.cdist``), optimal assignment of ALL rows to columns
  (``scipy.optimize.linear_sum_assignment``, the Crouse shortest-augmenting-path algorithm), then drop the
  pairs at or beyond the threshold.
* ``verifier.setup_match_blobs_roi`` (:122-160) and ``verifier.match_blobs_roi`` (:164-289): blobs of the
  inner part of an ROI are assigned first against all base blobs of the ROI, base blobs that stay unmatched
  then get a second chance against the blobs of the outer rim; matches are listed sorted by the base blob.
* ``colocalizer.colocalize_blobs_match`` (magmap/cv/colocalizer.py:444-501): every ordered pair of channels.
* ``colocalizer.StackColocalizer.colocalize_stack`` (:221-337): the stack is split again with a larger
  overlap, every block is matched on its own, and blobs matched in more than one block keep their shortest
  match (first of equals).

SciPy is the reference's own dependency here (``scipy.spatial`` / ``scipy.optimize``); the product replaces both
calls with its own code (``mmx_cdist_f64``, ``mmx_host_lsap``) and is tested against this file and against
fixtures from the real reference (``tests/golden/match.npz``).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
from scipy import optimize
from scipy.spatial import distance

from . import magmap_oracle as mmo


def find_closest_blobs_cdist(blobs, blobs_master, thresh=None, scaling=None):
    """verifier.py:47-119 -> ``(rows, cols, distances)``."""
    a, b = blobs, blobs_master
    if scaling is not None:
        n = len(scaling)
        a = np.multiply(blobs[:, :n], scaling)
        b = np.multiply(blobs_master[:, :n], scaling)
    dists = distance.cdist(a, b)
    rows, cols = optimize.linear_sum_assignment(dists)
    closest = dists[rows, cols]
    if thresh is not None:
        keep = closest < thresh
        rows, cols, closest = rows[keep], cols[keep], closest[keep]
    return rows, cols, closest


def get_blobs_in_roi(blobs, offset, size, margin=(0, 0, 0), reverse=True):
    """detector.py:1210-1245; ``offset`` / ``size`` arrive in x, y, z by default."""
    if reverse:
        offset, size, margin = offset[::-1], size[::-1], margin[::-1]
    mask = np.all([blobs[:, a] >= offset[a] - margin[a] for a in range(3)] +
                  [blobs[:, a] < offset[a] + size[a] + margin[a] for a in range(3)], axis=0)
    return blobs[mask], mask


def setup_match(tol):
    """verifier.py:122-160 without the ``resize_blobs`` branch: ``(thresh, scaling, inner_padding)``."""
    tol = np.asarray(tol, dtype=float)
    thresh = np.amax(tol)
    return thresh, thresh / tol, np.floor(tol[::-1])


def _match_list(blobs, blobs_master, close, close_master, dists):
    """verifier.py:23-44: ``(master, blob, distance)`` sorted by the master's z, y, x."""
    found_master = blobs_master[close_master]
    order = np.lexsort(tuple(found_master[:, i] for i in range(2, -1, -1)))
    return found_master[order], blobs[close][order], np.asarray(dists)[order]


def match_blobs_roi(blobs, blobs_base, offset, size, thresh, scaling, inner_padding):
    """verifier.py:164-289 -> ``(blob1 rows, blob2 rows, distances)`` (blob1 = base / master)."""
    size = np.asarray(size, dtype=float)
    pad_max = np.clip(np.ceil(np.divide(size, 2) - 1), 0, None)
    inner_padding = np.clip(inner_padding, 0, pad_max)
    size_inner = np.subtract(size, inner_padding * 2)
    offset_inner = np.add(offset, inner_padding)
    blobs_roi, _ = get_blobs_in_roi(blobs, offset, size)
    blobs_inner, inner_mask = get_blobs_in_roi(blobs_roi, offset_inner, size_inner)
    base_roi, _ = get_blobs_in_roi(blobs_base, offset, size)
    _, base_inner_mask = get_blobs_in_roi(base_roi, offset_inner, size_inner)
    found, found_base, dists = find_closest_blobs_cdist(blobs_inner, base_roi, thresh, scaling)
    blobs_inner[:, 4] = 0
    blobs_inner[found, 4] = 1
    base_roi[base_inner_mask, 5] = 0
    base_roi[found_base, 5] = 1
    missed = base_roi[base_roi[:, 5] == 0]
    outer = blobs_roi[np.invert(inner_mask)]
    found_out, found_base_out, dists_out = find_closest_blobs_cdist(outer, missed, thresh, scaling)
    missed[found_base_out, 5] = 1
    outer[found_out, 4] = 1
    m1 = _match_list(blobs_inner, base_roi, found, found_base, dists)
    m2 = _match_list(outer, missed, found_out, found_base_out, dists_out)
    return tuple(np.concatenate((a, b)) for a, b in zip(m1, m2))


def colocalize_blobs_match(table: np.ndarray, offset, size, tol, inner_padding=None, channels=None
                           ) -> Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """colocalizer.py:444-501 on the 8-column table -> ``{(chl, chl_other): (blob1, blob2, dist)}``; the
    confirmed / truth columns of the matched rows are reset to -1 (:496-497)."""
    thresh, scaling, inner_pad = setup_match(tol)
    if inner_padding is None:
        inner_padding = inner_pad
    out = {}
    chls = np.unique(table[:, 6]).astype(int)
    if channels is not None:
        chls = [c for c in chls if c in channels]
    for chl in chls:
        base = table[table[:, 6] == chl]
        for other in chls:
            if chl >= other:
                continue
            b1, b2, d = match_blobs_roi(table[table[:, 6] == other], base, np.asarray(offset), np.asarray(size),
                                        thresh, scaling, inner_padding)
            b1, b2 = b1.copy(), b2.copy()
            for b in (b1, b2):
                b[:, 4:6] = -1
            out[(int(chl), int(other))] = (b1, b2, d)
    return out


def colocalize_stack(shape, table: np.ndarray, profile: dict, resolutions, channels=None):
    """colocalizer.py:221-337 -> ``{(chl, chl_other): (blob1, blob2, dist)}`` after the de-duplication."""
    blocks = mmo.setup_blocks(profile, shape, resolutions)
    match_tol = np.multiply(blocks["overlap_base"], profile.get("verify_tol_factor", (1, 1, 1)))
    inner_pad = np.add(setup_match(match_tol)[2], blocks["overlap_base"])
    slices, offsets = mmo.stack_splitter(shape, blocks["max_pixels"], inner_pad[::-1])
    per_key: Dict[Tuple[int, int], list] = {}
    for coord in np.ndindex(*slices.shape):
        offset = offsets[coord]
        size = [s.stop - s.start for s in slices[coord]]
        got = colocalize_blobs_match(table, offset[::-1], size[::-1], match_tol, channels=channels)
        for key, val in got.items():
            per_key.setdefault(key, []).append(val)
    out = {}
    for key, parts in per_key.items():
        b1, b2, d = (np.concatenate([p[i] for p in parts]) for i in range(3))
        for which in (0, 1):
            cur = (b1, b2)[which]
            if not len(cur):
                continue
            _, first, inv, counts = np.unique(cur[:, :3], axis=0, return_index=True, return_inverse=True,
                                              return_counts=True)
            inv = np.asarray(inv).reshape(-1)
            if np.sum(counts > 1) > 0:
                keep = list(first[counts == 1])          # the singles, in np.unique order
                for i, ct in enumerate(counts):
                    if ct <= 1:
                        continue
                    rows = np.nonzero(inv == i)[0]
                    best = rows[d[rows] == np.amin(d[rows])]
                    keep.append(best[0])                 # first of equals
                keep = np.asarray(keep, dtype=int)
                b1, b2, d = b1[keep], b2[keep], d[keep]
        out[key] = (b1, b2, d)
    return out
