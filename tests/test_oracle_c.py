"""Pin the plain-C oracle (oracle/ndfilters.c) bit-for-bit against the installed SciPy.

CPU only.  SciPy 1.15.3 on this image is the reference's pinned SciPy
(envs/requirements.txt:47); its ``_nd_image`` C extension does the arithmetic of
``skimage.blob_log`` in the reference.  The C oracle fixes the exact operation
order that the device float64 re-score kernel reproduces.
"""
import numpy as np
import pytest
from scipy import ndimage as ndi
from scipy.ndimage import _filters as sci_filters

from conftest import load_golden
from oracle import blob_log_oracle as blo
from oracle import c_oracle


@pytest.mark.parametrize("sigma", [1.0, 1.7, 3.0, 3.5, 5.0])
def test_kernel_weights_equal_scipy(sigma):
    R = c_oracle.kernel_radius(sigma)
    assert R == int(4.0 * sigma + 0.5)
    for order in (0, 2):
        want = sci_filters._gaussian_kernel1d(sigma, order, R)[::-1]
        np.testing.assert_array_equal(c_oracle.gaussian_kernel1d(sigma, order, R), want)


@pytest.mark.parametrize("shape,sigma", [((20, 24, 28), 1.5), ((17, 33, 21), 3.0),
                                         ((5, 40, 9), 3.5), ((3, 4, 50), 5.0)])
def test_gaussian_laplace_bit_exact_f64(shape, sigma):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 65536, shape).astype(np.float64) * (1.0 / 65535)
    want = ndi.gaussian_laplace(img, sigma)
    got = c_oracle.gaussian_laplace(img, sigma)
    np.testing.assert_array_equal(got, want)


def test_gaussian_laplace_bit_exact_f32():
    rng = np.random.default_rng(2)
    img = rng.random((18, 22, 26)).astype(np.float32)
    np.testing.assert_array_equal(c_oracle.gaussian_laplace(img, 2.5), ndi.gaussian_laplace(img, 2.5))


def test_peak_mask_equals_numpy_oracle():
    g = load_golden("bloblog_u16_5sigma.npz")
    _, st = blo.blob_log(g["volume"], 3, 5, 5, 0.1, 0.5, return_stages=True)
    mask = c_oracle.peak_mask4d(st["cube"], 0.1)
    np.testing.assert_array_equal(mask, blo.peak_mask(st["cube"], 0.1))
    assert mask.sum() == len(g["peaks"])
    # constant cube: "trivial image" rule
    assert c_oracle.peak_mask4d(np.full((4, 5, 6, 2), 0.5), 0.1).sum() == 0
