"""CPU ORACLE (test infrastructure, NOT product code) -- intensity co-localisation restated.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.

``colocalizer.colocalize_blobs`` (reference magmap/cv/colocalizer.py:340-441) as called per block
from ``StackDetector.detect_sub_roi`` (magmap/cv/stack_detect.py:159-162) with the default
``thresh="min"``:

* per channel a label volume: -1 everywhere, the blob's row index at each blob centre of that
  channel, grey-dilated with ``morphology.ball(2)`` (33 voxels; scikit-image ->
  ``ndi.grey_dilation``, reflect) -- where two balls meet, the HIGHER row index owns the voxel;
* threshold of a channel = the minimum over its blobs of the mean intensity (in that channel) of
  the voxels the blob owns; a blob that owns nothing gives ``nan`` and poisons the threshold,
  exactly as NumPy does in the reference;
* flag ``[b, c] = 1`` iff the mean intensity of channel ``c`` over blob ``b``'s voxels ``>=``
  the threshold of ``c`` (channels without blobs are skipped).

Pinned by golden vectors from the real reference (``tests/golden/make_golden.py`` -> ``coloc.npz``,
``stack_coloc*.npz``).
"""
from __future__ import annotations

import warnings
from typing import Optional

import numpy as np
from scipy import ndimage as ndi


def ball(radius: int) -> np.ndarray:
    """``skimage.morphology.ball``: voxels with squared distance <= radius**2."""
    n = 2 * radius + 1
    z, y, x = np.mgrid[-radius:radius:n * 1j, -radius:radius:n * 1j, -radius:radius:n * 1j]
    return np.array((x * x + y * y + z * z) <= radius * radius, dtype=np.uint8)


def colocalize_blobs(roi: np.ndarray, blobs: Optional[np.ndarray], thresh=None) -> Optional[np.ndarray]:
    """``(len(blobs), n_channels)`` uint8 flags; colocalizer.py:340-441."""
    if blobs is None or roi is None or roi.ndim < 4:
        return None
    if thresh is None:
        thresh = "min"
    selem = ball(2)
    size = roi.shape[:3]
    in_roi = np.all([blobs[:, 0] >= 0, blobs[:, 0] < size[0], blobs[:, 1] >= 0, blobs[:, 1] < size[1],
                     blobs[:, 2] >= 0, blobs[:, 2] < size[2]], axis=0)       # get_blobs_in_roi
    blobs_roi = blobs[in_roi]
    blobs_chl = blobs_roi[:, 6]
    ranges, masks, threshs = [], [], []
    for chl in range(roi.shape[3]):
        sel = np.isin(blobs_chl, chl)
        rng = np.where(sel)[0]
        ranges.append(rng)
        mask = np.ones(size, dtype=int) * -1
        c = blobs_roi[sel, :3].astype(int)
        mask[c[:, 0], c[:, 1], c[:, 2]] = rng
        mask = ndi.grey_dilation(mask, footprint=selem)
        masks.append(mask)
        if thresh == "min":
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                threshs.append(None if len(rng) == 0 else np.amin(
                    [np.mean(roi[mask == b, chl]) for b in rng]))
        else:
            mb = mask >= 0
            threshs.append(np.percentile(roi if np.sum(mb) < 1 else roi[mb, chl], thresh))
    channels = np.unique(blobs_roi[:, 6]).astype(int)
    colocs_roi = np.zeros((blobs_roi.shape[0], roi.shape[3]), dtype=np.uint8)
    for chl in channels:
        mask = masks[chl]
        for other in channels:
            if threshs[other] is None:
                continue
            for b in ranges[chl]:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    avg = np.mean(roi[mask == b, other])
                if avg >= threshs[other]:
                    colocs_roi[b, other] = 1
    colocs = np.zeros((blobs.shape[0], roi.shape[3]), dtype=np.uint8)
    colocs[in_roi] = colocs_roi
    return colocs
