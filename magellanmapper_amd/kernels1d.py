"""Host-side filter parameters: Gaussian tap weights, radii and the sigma ladder.

Computed with NumPy on the host, operation for operation as SciPy / scikit-image do
on the same box, because the exact float64 re-score on the device must use the very
weights ``scipy.ndimage`` would have used:

* :func:`gaussian_half_kernel` -- ``scipy.ndimage._filters._gaussian_kernel1d``
  (scipy/ndimage/_filters.py:226-254) for orders 0 and 2, radius
  ``int(truncate * sigma + 0.5)`` (:313-315).  Only the half ``k = 0..R`` is kept: the
  kernels are exactly symmetric (checked).
* :func:`sigma_ladder` -- ``skimage.feature.blob_log``'s linear ladder
  (skimage/feature/blob.py:473-497) and its ``mean(sigma)**2`` scale factor (:501-502).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

TRUNCATE = 4.0


def kernel_radius(sigma: float) -> int:
    return int(TRUNCATE * float(sigma) + 0.5)


def gaussian_half_kernel(sigma: float, order: int, radius: int) -> np.ndarray:
    """Weights at distance ``k = 0..radius`` from the centre tap (float64)."""
    if order not in (0, 2):
        raise ValueError("only the plain and the second-derivative Gaussian are on this path")
    s2 = sigma * sigma
    pos = np.arange(-radius, radius + 1)
    bell = np.exp(-0.5 / s2 * pos ** 2)
    bell = bell / bell.sum()
    if order == 0:
        full = bell
    else:
        # q(x) for the 2nd derivative, built the way SciPy builds it: apply (D + P) twice to [1, 0, 0]
        powers = np.arange(3)
        coeff = np.zeros(3)
        coeff[0] = 1
        step = np.diag(powers[1:], 1) + np.diag(np.ones(2) / -s2, -1)
        for _ in range(2):
            coeff = step.dot(coeff)
        full = (pos[:, None] ** powers).dot(coeff) * bell
    full = full[::-1]  # correlate, not convolve (gaussian_filter1d reverses the kernel)
    left = full[radius::-1]
    right = full[radius:]
    if not np.array_equal(left, right):
        raise AssertionError("Gaussian kernel is not exactly symmetric")
    return np.ascontiguousarray(right, dtype=np.float64)


def sigma_ladder(min_sigma: float, max_sigma: float, num_sigma: int) -> Tuple[np.ndarray, np.ndarray]:
    """``(sigmas[num_sigma], norms[num_sigma])`` for scalar sigmas on a 3-D image.

    The ladder is computed on 3-vectors like scikit-image does and the scalar taken from
    column 0; ``norm = mean(row)**2`` keeps NumPy's rounding of the 3-element mean.
    """
    lo = np.full(3, min_sigma, dtype=float)
    hi = np.full(3, max_sigma, dtype=float)
    scale = np.linspace(0, 1, num_sigma)[:, np.newaxis]
    rows = scale * (hi - lo) + lo
    norms = np.array([np.mean(r) ** 2 for r in rows], dtype=np.float64)
    return np.ascontiguousarray(rows[:, 0]), norms
