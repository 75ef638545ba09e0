"""Image resampling between pyramid scales, on the GPU, bit-compatible with Pillow.

The reference resamples the image and the optimizer state with Pillow in 'F' mode
(``num_utils.resize``, num_utils.py:90-108; callers style_transfer.py:399-401 and
optimizers.py:53-61): separable convolution, horizontal pass then vertical pass, with
per-output-pixel windows ``[xmin, xmin + n)`` and normalised weights of a Lanczos-3 (or triangle)
kernel stretched by ``max(scale, 1)``; every pass accumulates in double and stores float32.
This module reproduces the coefficient tables on the host (a few KB) and lets
``stx_image_resample`` apply them to ``[C, H, W]`` device arrays.
"""

import ctypes
import functools
import math

import numpy as np

from . import lib

LANCZOS, BILINEAR = 'lanczos', 'bilinear'


def _sinc(x):
    return 1.0 if x == 0.0 else math.sin(x * math.pi) / (x * math.pi)


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def _triangle(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


_FILTERS = {LANCZOS: (_lanczos, 3.0), BILINEAR: (_triangle, 1.0)}


@functools.lru_cache(maxsize=64)
def coefficients(in_size, out_size, method):
    """(bounds int32 [out, 2] = (first input index, tap count), weights float64 [out, ksize]).
    Cached: a scale change resamples four arrays with the same two tables; treat as read-only."""
    filt, base_support = _FILTERS[method]
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    weights = np.zeros((out_size, ksize), np.float64)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        total = 0.0
        for v in w:
            total += v
        if total != 0.0:
            w = [v / total for v in w]
        weights[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    return bounds, weights


def resample_host(arr, hw, method=LANCZOS):
    """Numpy version of the same arithmetic (used by the tests as the bridge to Pillow)."""
    arr = np.asarray(arr, np.float32)
    c, h, w = arr.shape
    bx, kx = coefficients(w, hw[1], method)
    by, ky = coefficients(h, hw[0], method)
    tmp = np.empty((c, h, hw[1]), np.float32)
    for xx in range(hw[1]):
        x0, n = bx[xx]
        acc = np.zeros((c, h), np.float64)
        for x in range(n):
            acc += arr[:, :, x0 + x].astype(np.float64) * kx[xx, x]
        tmp[:, :, xx] = acc
    out = np.empty((c, hw[0], hw[1]), np.float32)
    for yy in range(hw[0]):
        y0, n = by[yy]
        acc = np.zeros((c, hw[1]), np.float64)
        for y in range(n):
            acc += tmp[:, y0 + y, :].astype(np.float64) * ky[yy, y]
        out[:, yy, :] = acc
    return out


def resample_device(engine, src, hw, method=LANCZOS, clamp_min_zero=False):
    """Resamples a DeviceArray [C, H, W] to [C, hw[0], hw[1]] on the GPU; returns a new array."""
    c, h, w = src.shape
    bx, kx = coefficients(w, hw[1], method)
    by, ky = coefficients(h, hw[0], method)
    dst = engine.empty((c, hw[0], hw[1]))
    lib.call('stx_image_resample', engine.handle, src.ptr, c, h, w, dst.ptr, int(hw[0]), int(hw[1]),
             bx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
             kx.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), kx.shape[1],
             by.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
             ky.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ky.shape[1],
             int(bool(clamp_min_zero)))
    return dst
