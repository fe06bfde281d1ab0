"""CPU ORACLE (test infrastructure, NOT product code) -- isotropic rescale ahead of detection.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.

``cv_nd.make_isotropic`` (reference magmap/cv/cv_nd.py:1040-1167) as ``detector.detect_blobs``
calls it when the profile's ``isotropic`` is set (magmap/cv/detector.py:893-897): the block is
resized to ``(shape * resolutions / min(resolutions) * isotropic).astype(int)`` with
``skimage.transform.resize(mode="reflect", preserve_range=True)`` (linear, clipped to the input range)
and cast back to the input dtype.

The arithmetic is scikit-image's, and it differs between releases: scikit-image >= 0.19 (the reference
pins 0.25.2) resizes every shape with ``scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True)``
after a Gaussian anti-aliasing filter along down-sampled axes; 0.18.3 (the only release that runs in
the build container) does the same through ``ndi.map_coordinates`` -- bit-identical, checked -- EXCEPT
for 3-D arrays whose last axis keeps its length, which it sends through its 2-D ``warp`` (x taken for
channels) with ~1e-10 differences.  This restatement follows the pinned release (``ndi.zoom``).  It is
pinned by golden vectors from the real reference for multichannel blocks and for blocks whose three axes
all change (the shapes for which 0.18.3 takes the same code path); for single-channel blocks rescaled
along z only it is the same SciPy call but NOT pinned by a fixture ("parity unpinned" for that shape).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi


def calc_isotropic_factor(scale, res) -> np.ndarray:
    """cv_nd.py:1040-1067."""
    resize_factor = np.divide(res, np.amin(res))
    resize_factor = resize_factor * scale
    return resize_factor


def resize(image: np.ndarray, output_shape, mode: str = "reflect") -> np.ndarray:
    """``skimage.transform.resize(image, output_shape, mode=mode, preserve_range=True)`` (>= 0.19):
    order 1, anti-aliasing when down-sampling, clipped to the input range; float result."""
    output_shape = tuple(int(v) for v in output_shape)
    ndi_mode = {"reflect": "mirror", "edge": "nearest"}[mode]
    img = image if image.dtype.char in "df" else image.astype(float)
    factors = np.divide(image.shape, output_shape)
    filtered = img
    if any(o < i for o, i in zip(output_shape, image.shape)):
        sigma = np.maximum(0, (factors - 1) / 2)
        filtered = ndi.gaussian_filter(img, sigma, cval=0, mode=ndi_mode)
    out = ndi.zoom(filtered, [1 / f for f in factors], order=1, mode=ndi_mode, cval=0, grid_mode=True)
    if out.shape != output_shape:
        raise AssertionError(f"zoom gave {out.shape}, wanted {output_shape}")
    np.clip(out, img.min(), img.max(), out=out)
    return out


def make_isotropic(roi: np.ndarray, scale, res) -> np.ndarray:
    """cv_nd.py:1070-1107 + ``rescale_resize`` (:1110-1164) for a target shape."""
    resize_factor = calc_isotropic_factor(scale, res)
    isotropic_shape = np.array(roi.shape)
    isotropic_shape[:3] = (isotropic_shape[:3] * resize_factor).astype(int)
    mode = "edge" if np.any(np.array(roi.shape) == 1) else "reflect"
    return resize(roi, isotropic_shape, mode).astype(roi.dtype)
