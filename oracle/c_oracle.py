"""CPU ORACLE (test infrastructure, NOT product code) -- ctypes loader for ``ndfilters.c``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmmoracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile ``ndfilters.c`` with gcc (seconds).  Returns the library path."""
    src = os.path.join(_HERE, "ndfilters.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        i64p = ctypes.POINTER(ctypes.c_int64)
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.mmo_correlate1d_f64.argtypes = [dp, dp, i64p, ctypes.c_int, dp, ctypes.c_int]
        _lib.mmo_correlate1d_f32.argtypes = [fp, fp, i64p, ctypes.c_int, dp, ctypes.c_int]
        _lib.mmo_gaussian_laplace_f64.argtypes = [dp, dp, i64p, dp, dp, ctypes.c_int]
        _lib.mmo_gaussian_laplace_f32.argtypes = [fp, fp, i64p, dp, dp, ctypes.c_int]
        _lib.mmo_peak_mask4d_f64.argtypes = [dp, i64p, ctypes.c_int, ctypes.c_double,
                                            ctypes.POINTER(ctypes.c_uint8)]
        _lib.mmo_peak_mask4d_f64.restype = ctypes.c_int64
    return _lib


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def gaussian_kernel1d(sigma: float, order: int, radius: int) -> np.ndarray:
    """SciPy's ``_gaussian_kernel1d`` (SCI/ndimage/_filters.py:226-254), reversed as
    ``gaussian_filter1d`` does (:321) -- restated, orders 0 and 2 only."""
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    if order == 0:
        return phi_x[::-1].copy()
    if order != 2:
        raise ValueError("only orders 0 and 2 are on the path")
    exponent_range = np.arange(order + 1)
    q = np.zeros(order + 1)
    q[0] = 1
    D = np.diag(exponent_range[1:], 1)
    P = np.diag(np.ones(order) / -sigma2, -1)
    Q_deriv = D + P
    for _ in range(order):
        q = Q_deriv.dot(q)
    q = (x[:, None] ** exponent_range).dot(q)
    return (q * phi_x)[::-1].copy()


def kernel_radius(sigma: float, truncate: float = 4.0) -> int:
    """``int(truncate * sigma + 0.5)``; SCI/ndimage/_filters.py:313-315."""
    return int(truncate * float(sigma) + 0.5)


def gaussian_laplace(image: np.ndarray, sigma: float) -> np.ndarray:
    """C restatement of ``scipy.ndimage.gaussian_laplace`` for a 3-D float32/float64 array."""
    image = np.ascontiguousarray(image)
    R = kernel_radius(sigma)
    w0 = gaussian_kernel1d(sigma, 0, R)
    w2 = gaussian_kernel1d(sigma, 2, R)
    dims = (ctypes.c_int64 * 3)(*image.shape)
    out = np.empty_like(image)
    dp = ctypes.c_double
    if image.dtype == np.float64:
        rc = lib().mmo_gaussian_laplace_f64(_ptr(image, dp), _ptr(out, dp), dims,
                                            _ptr(w0, dp), _ptr(w2, dp), R)
    elif image.dtype == np.float32:
        rc = lib().mmo_gaussian_laplace_f32(_ptr(image, ctypes.c_float), _ptr(out, ctypes.c_float),
                                            dims, _ptr(w0, dp), _ptr(w2, dp), R)
    else:
        raise TypeError(image.dtype)
    if rc != 0:
        raise MemoryError
    return out


def peak_mask4d(cube: np.ndarray, threshold: float) -> np.ndarray:
    cube = np.ascontiguousarray(cube, dtype=np.float64)
    dims = (ctypes.c_int64 * 3)(*cube.shape[:3])
    mask = np.zeros(cube.shape, dtype=np.uint8)
    lib().mmo_peak_mask4d_f64(_ptr(cube, ctypes.c_double), dims, cube.shape[3], float(threshold),
                              _ptr(mask, ctypes.c_uint8))
    return mask.astype(bool)
