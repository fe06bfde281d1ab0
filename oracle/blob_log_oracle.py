"""CPU ORACLE (test infrastructure, NOT product code) -- ``skimage.feature.blob_log`` restated.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package ``magellanmapper_amd`` never does.

What it restates
----------------
The arithmetic of the reference's hot path lives in a third-party dependency that
is not under ``/root/reference``: scikit-image (``envs/requirements.txt:46`` pins
0.25.2; the copy importable in the build container is **0.18.3**) calling SciPy
(``envs/requirements.txt:47`` pins 1.15.3 = the SciPy on this image).  The single
reference call site is ``magmap/cv/detector.py:931-933``.  This file restates the
published scikit-image 0.18.3 algorithm with NumPy + ``scipy.ndimage`` /
``scipy.spatial`` (SciPy *is* the reference's own pinned dependency and is the
code that does the arithmetic in the reference too):

=====================  =====================================================
here                   follows (SKI = skimage 0.18.3, SCI = scipy 1.15.3)
=====================  =====================================================
``img_as_float``       SKI/util/dtype.py:310-328 (``_convert``, uint -> float)
``sigma_list``         SKI/feature/blob.py:473-497
``log_cube``           SKI/feature/blob.py:499-504 -> SCI/ndimage/_filters.py:644-707
``peak_mask``          SKI/feature/peak.py:28-50, 66-76
``peak_coords``        SKI/feature/peak.py:9-25 (+ SKI/_shared/coord.py: no-op
                       ``ensure_spacing`` on integer coordinates at spacing 1)
``blob_overlap``       SKI/feature/blob.py:55-81, 84-143
``prune_blobs``        SKI/feature/blob.py:146-187
``blob_log``           SKI/feature/blob.py:378-535
=====================  =====================================================

Pinning: checked against golden vectors produced by the *real* scikit-image
0.18.3 ``blob_log`` and the real ``magmap.cv.detector.detect_blobs`` (imported
from ``/root/reference`` under ``/opt/conda/bin/python3.9``) by
``tests/golden/make_golden.py``; see ``tests/test_oracle_golden.py``.

Version skew (0.18.3 here vs the 0.25.2 pin) is listed in DESIGN.md.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage as ndi
from scipy import spatial


# --------------------------------------------------------------------------- A0
def img_as_float(image: np.ndarray) -> np.ndarray:
    """uint -> float64 as ``x * (1 / imax)`` (NOT ``x / imax``); floats pass through.

    SKI/util/dtype.py:310-328: unsigned ints go through
    ``np.multiply(image, 1. / imax_in, dtype=float64)``; signed ints through
    ``(image + 0.5) * (2 / (imax - imin))``; float16/32/64 are returned as-is.
    """
    image = np.asarray(image)
    kind = image.dtype.kind
    if kind == "f":
        return image
    if kind == "b":
        return image.astype(np.float64)
    if kind == "u":
        imax = np.iinfo(image.dtype).max
        return np.multiply(image, 1.0 / imax, dtype=np.float64)
    if kind == "i":
        info = np.iinfo(image.dtype)
        out = np.add(image, 0.5, dtype=np.float64)
        out *= 2 / (float(info.max) - float(info.min))
        return out
    raise ValueError(f"cannot convert {image.dtype} to float")


# --------------------------------------------------------------------------- A1
def sigma_list(min_sigma, max_sigma, num_sigma: int, ndim: int = 3) -> Tuple[np.ndarray, bool]:
    """Linear sigma ladder, one row per scale, one column per image axis.

    SKI/feature/blob.py:473-497 (``log_scale=False`` branch, the only one the
    reference uses, ``magmap/cv/detector.py:931-933``).
    Returns ``(sigmas[num_sigma, ndim], scalar_sigma)``.
    """
    scalar_sigma = bool(np.isscalar(max_sigma) and np.isscalar(min_sigma))
    if np.isscalar(max_sigma):
        max_sigma = np.full(ndim, max_sigma, dtype=float)
    if np.isscalar(min_sigma):
        min_sigma = np.full(ndim, min_sigma, dtype=float)
    min_sigma = np.asarray(min_sigma, dtype=float)
    max_sigma = np.asarray(max_sigma, dtype=float)
    scale = np.linspace(0, 1, num_sigma)[:, np.newaxis]
    return scale * (max_sigma - min_sigma) + min_sigma, scalar_sigma


# ------------------------------------------------------------------------ A2+A3
def log_cube(image_f: np.ndarray, sigmas: np.ndarray) -> np.ndarray:
    """Scale-normalised negative LoG stack ``(z, y, x, sigma)``.

    SKI/feature/blob.py:499-504: ``-gaussian_laplace(image, s) * mean(s)**2`` per
    scale, ``np.stack(axis=-1)``.  ``gaussian_laplace`` is
    SCI/ndimage/_filters.py:644-707: sum over axes of (2nd-derivative Gaussian on
    that axis, plain Gaussian on the others), every 1-D pass truncated at
    ``int(4*sigma + 0.5)`` with ``reflect`` boundaries (:226-254, :258-323).
    """
    planes = []
    for s in sigmas:
        norm = np.mean(s) ** 2
        if image_f.dtype != np.float64:
            # The fixtures come from NumPy 1.26, where ``float32_array * float64_scalar``
            # demotes the scalar and multiplies in float32 (value-based casting).  NumPy 2
            # (NEP 50) would promote to float64; scikit-image 0.25.2 stores into a float32
            # cube anyway.  Follow the pinned fixtures.
            norm = image_f.dtype.type(norm)
        planes.append(-ndi.gaussian_laplace(image_f, s) * norm)
    return np.stack(planes, axis=-1)


# --------------------------------------------------------------------------- A4
def peak_mask(cube: np.ndarray, threshold_abs: float, threshold_rel: Optional[float] = 0.0) -> np.ndarray:
    """Boolean mask of 3**ndim local maxima strictly above the threshold.

    SKI/feature/peak.py:66-76 (``_get_threshold``) and :28-50 (``_get_peak_mask``):
    ``maximum_filter(footprint=ones(3,...), mode='constant')`` (zero padded, also
    across the sigma axis ends), equality to the max keeps plateaus, a cube where
    every voxel equals its max has no peaks at all, then ``& (cube > threshold)``.
    """
    threshold = threshold_abs if threshold_abs is not None else cube.min()
    if threshold_rel is not None:
        threshold = max(threshold, threshold_rel * cube.max())
    if cube.size == 1:
        return cube > threshold
    footprint = np.ones((3,) * cube.ndim)
    cube_max = ndi.maximum_filter(cube, footprint=footprint, mode="constant")
    out = cube == cube_max
    if np.all(out):
        out[:] = False
    out &= cube > threshold
    return out


def peak_coords(cube: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """Peak coordinates, highest response first.

    SKI/feature/peak.py:9-25: ``np.nonzero`` (C order) then
    ``argsort(-intensities)`` (NumPy's default, unstable, sort).  The following
    ``ensure_spacing(spacing=1, p_norm=inf)`` (SKI/_shared/coord.py:5-95) only
    rejects points closer than 1 in Chebyshev distance, which cannot happen for
    distinct integer coordinates, so it is the identity here.
    """
    coord = np.nonzero(mask)
    intensities = cube[coord]
    order = np.argsort(-intensities)
    return np.transpose(coord)[order]


# --------------------------------------------------------------------------- A5
def _sphere_overlap(d: float, r1: float, r2: float) -> float:
    """SKI/feature/blob.py:55-81 (lens volume over the smaller sphere's volume)."""
    vol = (math.pi / (12 * d) * (r1 + r2 - d) ** 2 *
           (d ** 2 + 2 * d * (r1 + r2) - 3 * (r1 ** 2 + r2 ** 2) + 6 * r1 * r2))
    return vol / (4. / 3 * math.pi * min(r1, r2) ** 3)


def _disk_overlap(d: float, r1: float, r2: float) -> float:
    """SKI/feature/blob.py:18-52 (2-D case, kept for completeness)."""
    ratio1 = (d ** 2 + r1 ** 2 - r2 ** 2) / (2 * d * r1)
    ratio1 = np.clip(ratio1, -1, 1)
    acos1 = math.acos(ratio1)
    ratio2 = (d ** 2 + r2 ** 2 - r1 ** 2) / (2 * d * r2)
    ratio2 = np.clip(ratio2, -1, 1)
    acos2 = math.acos(ratio2)
    a = -d + r2 + r1
    b = d - r2 + r1
    c = d + r2 - r1
    d = d + r2 + r1
    area = (r1 ** 2 * acos1 + r2 ** 2 * acos2 - 0.5 * math.sqrt(abs(a * b * c * d)))
    return area / (math.pi * (min(r1, r2) ** 2))


def blob_overlap(blob1: np.ndarray, blob2: np.ndarray, sigma_dim: int = 1) -> float:
    """Overlap fraction of two blobs ``(coords..., sigma...)``; SKI/feature/blob.py:84-143."""
    ndim = len(blob1) - sigma_dim
    if ndim > 3:
        return 0.0
    root_ndim = math.sqrt(ndim)
    if blob1[-1] == blob2[-1] == 0:
        return 0.0
    elif blob1[-1] > blob2[-1]:
        max_sigma = blob1[-sigma_dim:]
        r1 = 1
        r2 = blob2[-1] / blob1[-1]
    else:
        max_sigma = blob2[-sigma_dim:]
        r2 = 1
        r1 = blob1[-1] / blob2[-1]
    pos1 = blob1[:ndim] / (max_sigma * root_ndim)
    pos2 = blob2[:ndim] / (max_sigma * root_ndim)
    d = np.sqrt(np.sum((pos2 - pos1) ** 2))
    if d > r1 + r2:
        return 0.0
    if d <= abs(r1 - r2):
        return 1.0
    if ndim == 2:
        return _disk_overlap(d, r1, r2)
    return _sphere_overlap(d, r1, r2)


def prune_blobs(blobs_array: np.ndarray, overlap: float, sigma_dim: int = 1,
                pair_order: Optional[np.ndarray] = None) -> np.ndarray:
    """Zero the smaller (on ties: the first) blob of every over-overlapping pair.

    SKI/feature/blob.py:146-187.  Pairs come from ``cKDTree.query_pairs`` within
    ``2 * sigma_max * sqrt(ndim)`` and are visited in the iteration order of the
    returned Python ``set``; sigmas are zeroed in place, so the outcome can depend
    on that order when over-overlapping pairs share a blob.  ``pair_order`` lets a
    test substitute another visiting order (order-invariance checks).
    """
    blobs_array = np.array(blobs_array, dtype=np.float64, copy=True)
    sigma = blobs_array[:, -sigma_dim:].max()
    distance = 2 * sigma * math.sqrt(blobs_array.shape[1] - sigma_dim)
    tree = spatial.cKDTree(blobs_array[:, :-sigma_dim])
    pairs = np.array(list(tree.query_pairs(distance)))
    if len(pairs) == 0:
        return blobs_array
    if pair_order is not None:
        pairs = pairs[pair_order]
    for (i, j) in pairs:
        blob1, blob2 = blobs_array[i], blobs_array[j]
        if blob_overlap(blob1, blob2, sigma_dim=sigma_dim) > overlap:
            if blob1[-1] > blob2[-1]:
                blob2[-1] = 0
            else:
                blob1[-1] = 0
    return np.stack([b for b in blobs_array if b[-1] > 0])


# ------------------------------------------------------------------------ driver
def blob_log(image: np.ndarray, min_sigma=1, max_sigma=50, num_sigma: int = 10,
             threshold: float = .2, overlap: float = .5, *, return_stages: bool = False):
    """``skimage.feature.blob_log`` (0.18.3) for ``log_scale=False, exclude_border=False``.

    SKI/feature/blob.py:378-535.  Returns ``(n, ndim + 1)`` float64 rows
    ``(coords..., sigma)`` for scalar sigmas; ``np.empty((0, 3))`` when there are
    no peaks (:516-517).  With ``return_stages`` also returns a dict holding the
    sigma ladder, the cube and the ordered raw peaks (used to build fixtures).
    """
    image_f = img_as_float(image)
    sigmas, scalar_sigma = sigma_list(min_sigma, max_sigma, num_sigma, image_f.ndim)
    cube = log_cube(image_f, sigmas)
    mask = peak_mask(cube, threshold, 0.0)
    local_maxima = peak_coords(cube, mask)
    stages = None
    if return_stages:
        stages = {"sigmas": sigmas, "cube": cube, "peaks": local_maxima,
                  "peak_values": cube[tuple(local_maxima.T)] if local_maxima.size else np.empty(0)}
    if local_maxima.size == 0:
        out = np.empty((0, 3))
        return (out, stages) if return_stages else out
    lm = local_maxima.astype(np.float64)
    sigmas_of_peaks = sigmas[local_maxima[:, -1]]
    if scalar_sigma:
        sigmas_of_peaks = sigmas_of_peaks[:, 0:1]
    lm = np.hstack([lm[:, :-1], sigmas_of_peaks])
    out = prune_blobs(lm, overlap, sigma_dim=sigmas_of_peaks.shape[1])
    return (out, stages) if return_stages else out
