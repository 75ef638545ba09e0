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
