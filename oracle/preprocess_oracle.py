"""CPU ORACLE (test infrastructure, NOT product code) -- per-block preprocessing restated.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.

What the reference does to every block before ``blob_log`` when the profile's
``denoise_size`` is set (it is, by default: ``magmap/settings/roi_prof.py:119``):
``magmap/cv/stack_detect.py:122-150`` splits the block into ``denoise_max_shape`` sub-blocks
(no overlap), runs ``plot_3d.saturate_roi`` (``magmap/plot/plot_3d.py:55-112``) and
``plot_3d.denoise_roi`` (:115-172) on each and re-merges them into one float64 block.
The two functions lean on scikit-image (0.18.3 here) which in turn calls SciPy:

* ``filters.gaussian(img, 8)`` -> ``ndi.gaussian_filter(float image, 8, mode='nearest',
  truncate=4)``; a 3-D array whose last axis has length 3 is taken for RGB and NOT blurred
  along that axis (skimage/filters/_gaussian.py:105-126);
* ``morphology.erosion(img, octahedron(1))`` -> ``ndi.grey_erosion(img, footprint=...)``
  (default ``reflect`` mode; skimage/morphology/grey.py:181-186).

Pinned by golden vectors from the real reference (``tests/golden/make_golden.py`` ->
``preproc*.npz``; ``tests/test_oracle_golden.py``).  ``tot_var_denoise`` (TV-Chambolle, off in
every stock nuclei profile except ``2p20x`` / ``minpreproc``) is not restated.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
from scipy import ndimage as ndi

#: ``morphology.octahedron(1)``: the 6-neighbourhood plus the centre
OCTAHEDRON1 = np.zeros((3, 3, 3), dtype=np.uint8)
OCTAHEDRON1[1, 1, :] = OCTAHEDRON1[1, :, 1] = OCTAHEDRON1[:, 1, 1] = 1


def _channels(roi, channel):
    multichannel = roi.ndim > 3
    if not multichannel:
        return False, [0]
    return True, (range(roi.shape[3]) if channel is None else channel)


def _prof(profiles, chl):
    return profiles[chl] if len(profiles) > chl else profiles[0]


def saturate_roi(roi: np.ndarray, profiles: Sequence[dict], near_max: Sequence[float],
                 channel: Optional[Sequence[int]] = None) -> np.ndarray:
    """Percentile contrast stretch to 0-1; magmap/plot/plot_3d.py:55-112."""
    multichannel, channels = _channels(roi, channel)
    roi_out = None
    for chl in channels:
        roi_show = roi[..., chl] if multichannel else roi
        settings = _prof(profiles, chl)
        vmin, vmax = np.percentile(roi_show, (settings["clip_vmin"], settings["clip_vmax"]))
        if vmin == vmax:
            saturated = roi_show
        else:
            max_thresh = near_max[chl] * settings["max_thresh_factor"]
            if vmax < max_thresh:
                vmax = max_thresh
            saturated = np.clip(roi_show, vmin, vmax)
            saturated = (saturated - vmin) / (vmax - vmin)
        if multichannel:
            if roi_out is None:
                roi_out = np.zeros(roi.shape, dtype=saturated.dtype)
            roi_out[..., chl] = saturated
        else:
            roi_out = saturated
    return roi_out


#: test hook: the full (2R+1) sigma-8 kernel to use instead of SciPy's.  ``np.exp`` differs by an ulp
#: between NumPy releases, so the golden fixtures (made under NumPy 1.26) carry the weights they used.
GAUSS_WEIGHTS: Optional[np.ndarray] = None


def gaussian(image: np.ndarray, sigma: float) -> np.ndarray:
    """``skimage.filters.gaussian(image, sigma)`` (0.18.3 defaults) for a float image."""
    sig = [sigma] * image.ndim
    if image.ndim == 3 and image.shape[-1] == 3:
        sig[-1] = 0                      # "(M, N, 3) is interpreted as 2D+RGB by default"
    image = image if image.dtype.kind == "f" else image.astype(np.float64)
    out = np.empty_like(image)
    if GAUSS_WEIGHTS is None:
        ndi.gaussian_filter(image, sig, output=out, mode="nearest", cval=0, truncate=4.0)
        return out
    # scipy/ndimage/_filters.py gaussian_filter: one correlate1d per axis with sigma > 1e-15, in axis
    # order, the first from the input and the rest in place
    src = image
    for axis, s in enumerate(sig):
        if s > 1e-15:
            ndi.correlate1d(src, GAUSS_WEIGHTS[::-1], axis, out, "nearest", 0.0, 0)
            src = out
    if src is image:
        out[...] = image
    return out


def denoise_tv_chambolle(image: np.ndarray, weight=0.1, eps=2.e-4, n_iter_max=200) -> np.ndarray:
    """``skimage.restoration.denoise_tv_chambolle`` for one float n-D image (0.18.3
    restoration/_denoise.py:315-393; unchanged in the arithmetic through 0.25): Chambolle's projection
    algorithm on the dual field ``p``, stopped when the energy changes by less than ``eps * E_0``.
    Every operation in NumPy's order -- the device reproduces the result bit for bit, including the
    iteration at which it stops, so the energies are summed with ``ndarray.sum`` (pairwise) here too."""
    ndim = image.ndim
    p = np.zeros((ndim,) + image.shape, dtype=image.dtype)
    g = np.zeros_like(p)
    d = np.zeros_like(image)
    out = image
    E_init = E_previous = 0.0
    i = 0
    while i < n_iter_max:
        if i > 0:
            d = -p.sum(0)                              # minus the divergence of p
            for ax in range(ndim):
                hi = [slice(None)] * ndim
                lo = [slice(None)] * ndim
                hi[ax], lo[ax] = slice(1, None), slice(0, -1)
                d[tuple(hi)] += p[ax][tuple(lo)]
            out = image + d
        else:
            out = image
        E = (d ** 2).sum()
        for ax in range(ndim):
            sl = [slice(None)] * ndim
            sl[ax] = slice(0, -1)
            g[ax][tuple(sl)] = np.diff(out, axis=ax)    # forward differences; the last index stays 0
        norm = np.sqrt((g ** 2).sum(axis=0))[np.newaxis, ...]
        E += weight * norm.sum()
        tau = 1. / (2. * ndim)
        norm *= tau / weight
        norm += 1.
        p -= tau * g
        p /= norm
        E /= float(image.size)
        if i == 0:
            E_init = E
            E_previous = E
        else:
            if np.abs(E_previous - E) < eps * E_init:
                break
            E_previous = E
        i += 1
    return out


def denoise_roi(roi: np.ndarray, profiles: Sequence[dict],
                channel: Optional[Sequence[int]] = None) -> np.ndarray:
    """Clip, unsharp mask (sigma 8) and density-dependent erosion; plot_3d.py:115-172."""
    multichannel, channels = _channels(roi, channel)
    roi_out = None
    for chl in channels:
        roi_show = roi[..., chl] if multichannel else roi
        settings = _prof(profiles, chl)
        saturated_mean = np.mean(roi_show)
        denoised = np.clip(roi_show, settings["clip_min"], settings["clip_max"])
        tot_var_denoise = settings.get("tot_var_denoise")
        if tot_var_denoise:
            denoised = denoise_tv_chambolle(denoised, weight=tot_var_denoise)
        unsharp_strength = settings["unsharp_strength"]
        if unsharp_strength:
            blurred = gaussian(denoised, 8)
            high_pass = denoised - unsharp_strength * blurred
            denoised = denoised + high_pass
        thresh_eros = settings["erosion_threshold"]
        if thresh_eros and saturated_mean > thresh_eros:
            out = np.empty_like(denoised)
            ndi.grey_erosion(denoised, footprint=OCTAHEDRON1, output=out)
            denoised = out
        if multichannel:
            if roi_out is None:
                roi_out = np.zeros(roi.shape, dtype=denoised.dtype)
            roi_out[..., chl] = denoised
        else:
            roi_out = denoised
    return roi_out


def preprocess_block(sub_roi: np.ndarray, denoise_max_shape, profiles: Sequence[dict],
                     near_max: Sequence[float]) -> np.ndarray:
    """The sub-sub-block loop of ``detect_sub_roi`` (stack_detect.py:122-150)."""
    shape3 = np.asarray(sub_roi.shape[:3])
    dms = np.asarray(denoise_max_shape)
    grid = (-(-shape3 // dms)).astype(int)
    merged = None
    for c in np.ndindex(*grid):
        sl = tuple(slice(int(c[a] * dms[a]), int(min((c[a] + 1) * dms[a], shape3[a]))) for a in range(3))
        piece = denoise_roi(saturate_roi(sub_roi[sl], profiles, near_max), profiles)
        if merged is None:
            merged = np.zeros(sub_roi.shape, dtype=piece.dtype)
        merged[sl] = piece
    return merged
