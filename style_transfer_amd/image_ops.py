"""Device-resident full-image operations (thin wrappers over the stx_image_* / stx_vec_* ABI).

These replace the host-side numpy of the reference's step loop: ``roll2`` + tile slicing
(``num_utils.py:136-140``, ``style_transfer.py:632,642``), ``tv_norm`` / ``p_norm`` / aux term
(``style_transfer.py:709-733``), the statistics of ``transfer`` (``style_transfer.py:808-815``)
and ``get_image`` (``style_transfer.py:378-386``).  Images are ``DeviceArray`` [3,H,W] on the
engine's GPU and stay UN-rolled: the per-iteration shift is an index offset in cut / put.
"""

import ctypes

import numpy as np

from . import lib


def _xy(roll):
    if roll is None:
        return (ctypes.c_int * 2)(0, 0)
    return (ctypes.c_int * 2)(int(roll[0]), int(roll[1]))


def cut_tile(engine, img, roll_xy, rect, tile):
    """tile <- window rect=(y0,y1,x0,x1) of roll2(img, roll_xy)."""
    y0, y1, x0, x1 = rect
    _, H, W = img.shape
    lib.call('stx_image_cut_tile', engine.handle, img.ptr, H, W, _xy(roll_xy), y0, x0, y1 - y0,
             x1 - x0, tile.ptr)


def put_tile(engine, grad, roll_xy, rect, tile_grad):
    """Places a tile gradient back into the un-rolled full gradient."""
    y0, y1, x0, x1 = rect
    _, H, W = grad.shape
    lib.call('stx_image_put_tile', engine.handle, grad.ptr, H, W, _xy(roll_xy), y0, x0, y1 - y0,
             x1 - x0, tile_grad.ptr)


class PendingScalar:
    def __init__(self):
        self._v = ctypes.c_double(float('nan'))

    @property
    def value(self):
        return self._v.value


def regularizers(engine, img, grad, mean_bgr, tv_scale, tv_power, p_scale, p_power, aux=None,
                 aux_scale=0.0, aux_roll=None):
    """grad += regularizer gradients; returns a PendingScalar with the loss (valid after sync).
    ``aux_roll``: the iteration's shift (x, y) -- the reference rolls the image, not the auxiliary
    image, so the un-rolled image meets the auxiliary image displaced by it."""
    _, H, W = img.shape
    mean = (ctypes.c_float * 3)(*[float(m) for m in np.ravel(mean_bgr)])
    out = engine.keep_until_sync(PendingScalar())
    lib.call('stx_image_regularizers', engine.handle, img.ptr, grad.ptr, H, W, mean,
             float(tv_scale), float(tv_power), float(p_scale), float(p_power),
             aux.ptr if aux is not None else None, float(aux_scale),
             _xy(aux_roll) if aux_roll is not None else None, ctypes.byref(out._v))
    return out


def swt_haar(engine, img, grad, scale, power, roll=None):
    """grad += scale * (p-norm gradient at the Haar SWT detail image of the rolled picture / 127.5);
    returns a PendingScalar with scale * sum |detail|^power (style_transfer.py:716-720 for the
    default wavelet and level count)."""
    _, H, W = img.shape
    out = engine.keep_until_sync(PendingScalar())
    lib.call('stx_image_swt_haar', engine.handle, img.ptr, grad.ptr, H, W,
             _xy(roll) if roll is not None else None, float(scale), float(power),
             ctypes.byref(out._v))
    return out


def adam_step(engine, params, grad, g1, g2, p1, avg, lr, b1, b2, bp1, corr1, corr2, corrp):
    lib.call('stx_adam_step', engine.handle, params.ptr, grad.ptr, g1.ptr, g2.ptr, p1.ptr, avg.ptr,
             params.size, float(lr), float(b1), float(b2), float(bp1), float(corr1), float(corr2),
             float(corrp))


def dot(engine, x, y):
    out = ctypes.c_double(0)
    lib.call('stx_vec_dot', engine.handle, x.ptr, y.ptr, x.size, ctypes.byref(out))
    return out.value


def mean_abs(engine, x):
    out = ctypes.c_double(0)
    lib.call('stx_vec_mean_abs', engine.handle, x.ptr, x.size, ctypes.byref(out))
    return out.value


def axpy(engine, a, x, y):
    lib.call('stx_vec_axpy', engine.handle, float(a), x.ptr, y.ptr, x.size)


class DeviceScalars:
    """A few float64 slots on the engine's GPU for scalars that never need to visit the host."""

    def __init__(self, engine, n):
        self.engine, self.n = engine, n
        self.array = engine.empty((n,), np.float64)

    def ptr(self, i):
        assert 0 <= i < self.n
        return self.array.ptr + 8 * i

    def free(self):
        self.array.free()


def dot_async(engine, x, y, out_ptr):
    """*out_ptr (device double) = <x, y>; no synchronisation."""
    lib.call('stx_vec_dot_async', engine.handle, x.ptr, y.ptr, x.size, out_ptr)


def abs_sum_async(engine, x, out_ptr):
    lib.call('stx_vec_abs_sum_async', engine.handle, x.ptr, x.size, out_ptr)


def axpy_dev(engine, c1, a_ptr, da, x, y, c2=0.0, b_ptr=None, db=1.0):
    """y += float(*a / da * c1 [+ *b / db * c2]) * x with a, b device doubles."""
    lib.call('stx_vec_axpy_dev', engine.handle, float(c1), a_ptr, float(da), float(c2), b_ptr,
             float(db), x.ptr, y.ptr, x.size)


def scale_dev(engine, c, den_ptr, x, den_div=1.0):
    """x *= float(c / (*den / den_div)) with den a device double."""
    lib.call('stx_vec_scale_dev', engine.handle, float(c), den_ptr, float(den_div), x.ptr, x.size)


def axpy_dot_dev(engine, c1, a_ptr, da, x, y, z, out_ptr, c2=0.0, b_ptr=None, db=1.0, src=None,
                 scale_c=0.0, scale_den_ptr=None, scale_div=1.0):
    """y = coef * x + src (coef as axpy_dev; src defaults to y), then -- with scale_den_ptr -- y *=
    float(scale_c / (*scale_den / scale_div)); *out_ptr = <z, y>: the axpy of one iteration of the
    two-loop recursion (and the scaling between the loops) with the dot product of the next, one pass."""
    lib.call('stx_vec_axpy_dot_dev', engine.handle, float(c1), a_ptr, float(da), float(c2), b_ptr,
             float(db), float(scale_c), scale_den_ptr, float(scale_div), x.ptr,
             (y if src is None else src).ptr, y.ptr, z.ptr, x.size, out_ptr)


def lbfgs_pair(engine, g_new, g_old, s, y, out_ptr2):
    """y = g_new - g_old, g_old = g_new, out_ptr2[0] = <s, y>, out_ptr2[1] = <y, y>; returns <s, y>
    (one host synchronisation)."""
    sy = ctypes.c_double()
    lib.call('stx_vec_lbfgs_pair', engine.handle, g_new.ptr, g_old.ptr, s.ptr, y.ptr, s.size, out_ptr2,
             ctypes.byref(sy))
    return sy.value


def scale2_axpy(engine, c1, c2, s, params):
    """s = c2 * (c1 * s); params += s."""
    lib.call('stx_vec_scale2_axpy', engine.handle, float(c1), float(c2), s.ptr, params.ptr, s.size)


def scale(engine, a, x):
    lib.call('stx_vec_scale', engine.handle, float(a), x.ptr, x.size)


def step_stats(engine, avg, old):
    """(mean|avg-old|, sqrt(mean(xdiff^2+ydiff^2))); old <- avg."""
    _, H, W = avg.shape
    out = (ctypes.c_double * 2)()
    lib.call('stx_image_step_stats', engine.handle, avg.ptr, old.ptr, H, W, out)
    return out[0], out[1]


class PendingStats:
    """step_stats whose two sums are still on their way: ``values()`` after the engine's pending
    values were published (sync, or wait_fence on a later fence)."""

    def __init__(self, n):
        self._raw = (ctypes.c_double * 2)(float('nan'), float('nan'))
        self._n = n

    def values(self):
        return self._raw[0] / self._n, float(np.sqrt(self._raw[1] / self._n))


def step_stats_async(engine, avg, old):
    """step_stats without the host wait (stx_image_step_stats_async); old <- avg in stream order."""
    _, H, W = avg.shape
    out = engine.keep_until_sync(PendingStats(3.0 * H * W))
    lib.call('stx_image_step_stats_async', engine.handle, avg.ptr, old.ptr, H, W, out._raw)
    return out


def to_u8(engine, img, mean_bgr):
    """RGB HWC uint8 ndarray of img + mean, clipped and truncated like the reference."""
    _, H, W = img.shape
    mean = (ctypes.c_float * 3)(*[float(m) for m in np.ravel(mean_bgr)])
    out = engine.empty((H, W, 3), np.uint8)
    lib.call('stx_image_to_u8', engine.handle, img.ptr, H, W, mean, out.ptr)
    host = out.get()
    out.free()
    return host
