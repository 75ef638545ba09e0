"""Numpy restatement of the reference's numeric helpers (``num_utils.py``).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference routes these through
scipy's float32 BLAS (``num_utils.py:20-66``); plain float32 numpy is used here, which
differs only in summation order.  Each function cites the lines it follows.
"""

import numpy as np

EPS = np.finfo(np.float32).eps          # num_utils.py:14


def half_sq_norm(x):
    """1/2 * <x, x>  (num_utils.py:69-71 ``norm2``)."""
    v = np.asarray(x, np.float32).ravel()
    return float(np.dot(v, v)) / 2


def l1_normalize(x):
    """Scales x IN PLACE so that mean|x| == 1 (num_utils.py:85-87 ``normalize``)."""
    x *= np.float32(1 / (float(np.abs(x).sum(dtype=np.float32)) / x.size + EPS))
    return x


def gram_lower(feat):
    """Lower triangle of F F^T / F.size, upper triangle zero (num_utils.py:143-147 + 53-56)."""
    c = feat.shape[0]
    f = feat.reshape(c, -1)
    return np.tril((f @ f.T) * np.float32(1 / f.size)).astype(np.float32)


def symm_lower_times(a_lower, b):
    """sym(tril(a)) @ b, reading only a's lower triangle (num_utils.py:60-66 ``ssymm``)."""
    full = np.tril(a_lower) + np.tril(a_lower, -1).T
    return (full.astype(np.float32) @ b).astype(np.float32)


def roll_xy(arr, xy):
    """Circular shift: xy[0] moves the LAST axis, xy[1] the one before (num_utils.py:136-140)."""
    if xy is None or not np.any(np.asarray(xy) != 0):
        return arr
    arr[...] = np.roll(arr, (int(xy[0]), int(xy[1])), axis=(-1, -2))
    return arr


def tv_loss_grad(x, beta=2):
    """Total-variation norm with circular forward differences (num_utils.py:150-162)."""
    x = np.asarray(x, np.float32)
    dx = x - np.roll(x, -1, axis=2)
    dy = x - np.roll(x, -1, axis=1)
    n2 = dx * dx + dy * dy + EPS
    loss = float(np.sum(n2 ** np.float32(beta / 2)))
    dn = np.float32(beta / 2) * n2 ** np.float32(beta / 2 - 1)
    gx = 2 * dx * dn
    gy = 2 * dy * dn
    grad = gx + gy - np.roll(gx, 1, axis=2) - np.roll(gy, 1, axis=1)
    return loss, grad.astype(np.float32)


def p_norm_loss_grad(x, p=2):
    """sum |x|^p and its gradient (num_utils.py:74-82)."""
    x = np.asarray(x, np.float32)
    if p == 1:
        return float(np.abs(x).sum(dtype=np.float32)), np.sign(x)
    if p == 2:
        return float(np.dot(x.ravel(), x.ravel())), 2 * x
    a = np.abs(x)
    a1 = a ** np.float32(p - 1)
    return float(np.dot(a1.ravel(), a.ravel())), (np.float32(p) * np.sign(x) * a1).astype(np.float32)


# ------------------------------------------------------------------------------------------- SWT
# num_utils.py:165-196: swt_norm(x, wavelet, level, p) pads every channel symmetrically to a square
# of side 2**ceil(log2(max(H, W))), takes pywt.swt2, zeroes the approximation band, inverts with
# pywt.iswt2, crops, and returns p_norm of the result (the gradient is the p-norm's gradient at
# the detail image, NOT chained through the transform).  PyWavelets (pinned by nothing upstream:
# `import pywt` at num_utils.py:9) is absent from /root/reference and from this image, so no
# reference vectors exist: PARITY UNPINNED.  Restated for the defaults of the reference's command
# line only ('haar', 1 level; config_system.py:84-87):
#   * one level of the stationary (undecimated) Haar transform with periodic extension:
#       a[n] = (x[n] + x[n+1]) / sqrt 2,  d[n] = (x[n] - x[n+1]) / sqrt 2   (every shift n)
#   * its inverse averages the two reconstructions that a decimated transform would give from the
#     even and from the odd shifts; with the approximation band zeroed what remains is the detail
#     projector  x[n] - (x[n-1] + 2 x[n] + x[n+1]) / 4  per axis.  In 2-D only the low-low band is
#     zeroed, so the detail image is  x - B x  with B = [1 2 1]/4 (rows) x [1 2 1]/4 (columns),
#     circular on the padded square.
# swt_haar1_filterbank below does the transform and its inverse band by band (no closed form) and
# tests/test_oracle_swt.py holds the two against each other.

def _pad_width(shape, divisors):
    """num_utils.py:165-176."""
    pw = []
    for length, divisor in zip(shape, divisors):
        to_pad = int(np.ceil(length / divisor)) * divisor - length
        pw.append((to_pad // 2, to_pad // 2) if to_pad % 2 == 0 else (to_pad // 2, to_pad // 2 + 1))
    return pw


def swt_haar1_detail(x):
    """Detail part (approximation band zeroed) of the one-level stationary Haar transform of every
    channel of x [C,H,W], computed on the symmetric padding to a power-of-two square and cropped
    back (num_utils.py:184-196 with wavelet='haar', level=1)."""
    x = np.asarray(x, np.float32)
    div = 2 ** int(np.ceil(np.log2(max(x.shape[1:]))))
    pw = _pad_width(x.shape, (1, div, div))
    xp = np.pad(x, pw, 'symmetric')
    blur = xp
    for axis in (1, 2):
        blur = (np.roll(blur, 1, axis) + 2 * blur + np.roll(blur, -1, axis)) / np.float32(4)
    d = xp - blur
    return d[:, pw[1][0]:pw[1][0] + x.shape[1], pw[2][0]:pw[2][0] + x.shape[2]].astype(np.float32)


def swt_haar1_filterbank(ch):
    """The same for ONE square 2-D array, band by band: undecimated Haar analysis along both
    axes, low-low band dropped, synthesis as the average over the shifts (float64)."""
    ch = np.asarray(ch, np.float64)
    s = np.sqrt(0.5)

    def analysis(a, axis):
        nxt = np.roll(a, -1, axis)
        return s * (a + nxt), s * (a - nxt)

    def synthesis(lo, hi, axis):
        # x[n] from (lo[n], hi[n]) = s (x[n] +- x[n+1]) and from (lo[n-1], hi[n-1]); mean of both
        from_n = s * (lo + hi)
        from_prev = s * (np.roll(lo, 1, axis) - np.roll(hi, 1, axis))
        return 0.5 * (from_n + from_prev)

    lo, hi = analysis(ch, 0)
    ll, lh = analysis(lo, 1)
    hl, hh = analysis(hi, 1)
    ll = np.zeros_like(ll)                       # coeffs[0][0][...] = 0  (num_utils.py:191-192)
    lo_r = synthesis(ll, lh, 1)
    hi_r = synthesis(hl, hh, 1)
    return synthesis(lo_r, hi_r, 0)


def swt_norm_haar1(x, p=2):
    """(loss, grad) of num_utils.swt_norm(x, 'haar', 1, p)."""
    return p_norm_loss_grad(swt_haar1_detail(x), p)
