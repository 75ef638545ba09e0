"""Numpy restatement of the BVLC/caffe layer arithmetic used by the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference never implements these
itself: it calls pycaffe (``style_transfer.py:370,425,566,608-610``) on the layer
graph in ``vgg19.prototxt``.  Semantics restated here (Caffe, documented behaviour):

* Convolution: cross-correlation (no kernel flip), weights ``[Cout, Cin, kh, kw]``,
  zero padding, stride 1, plus bias.  Backward-to-data overwrites the bottom diff.
* ReLU (in-place, ``vgg19.prototxt:27-32``): ``max(0, x)``; backward multiplies the
  diff by ``data > 0`` evaluated on the (already rectified) blob.
* Pooling (``vgg19.prototxt:50-60``): 2x2 stride 2, output ``ceil((H-k)/s)+1``;
  MAX keeps the first maximum in row-major window order (strict ``>`` scan);
  AVE divides by the window size clipped to the blob.

All arithmetic is float32, like Caffe's CPU path (im2col + SGEMM).
"""

import numpy as np


def conv_forward(x, w, b, pad=1):
    """x [Cin,H,W] f32, w [Cout,Cin,kh,kw], b [Cout] -> [Cout,H',W'] (stride 1)."""
    cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    oh, ow = h + 2 * pad - kh + 1, wd + 2 * pad - kw + 1
    xp = np.zeros((cin, h + 2 * pad, wd + 2 * pad), np.float32)
    xp[:, pad:pad + h, pad:pad + wd] = x
    cols = np.empty((cin, kh, kw, oh, ow), np.float32)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, ky, kx] = xp[:, ky:ky + oh, kx:kx + ow]
    y = w.reshape(cout, -1) @ cols.reshape(cin * kh * kw, oh * ow)
    y += b.astype(np.float32)[:, None]
    return y.reshape(cout, oh, ow)


def conv_backward_data(dy, w, pad=1):
    """Gradient w.r.t. the conv input: dy [Cout,H',W'] -> [Cin,H,W] (col2im of W^T dy)."""
    cout, cin, kh, kw = w.shape
    _, oh, ow = dy.shape
    h, wd = oh - 2 * pad + kh - 1, ow - 2 * pad + kw - 1
    cols = (w.reshape(cout, -1).T @ dy.reshape(cout, oh * ow)).reshape(cin, kh, kw, oh, ow)
    dxp = np.zeros((cin, h + 2 * pad, wd + 2 * pad), np.float32)
    for ky in range(kh):
        for kx in range(kw):
            dxp[:, ky:ky + oh, kx:kx + ow] += cols[:, ky, kx]
    return np.ascontiguousarray(dxp[:, pad:pad + h, pad:pad + wd])


def pooled_size(n, k=2, s=2):
    """Caffe's ceil-mode output length (pad 0)."""
    return int(np.ceil((n - k) / s)) + 1 if n >= k else 1


def _windows(x, k=2, s=2):
    """Returns (stack [k*k, C, oh, ow] padded with -inf/0 mask, valid mask [k*k, oh, ow])."""
    c, h, w = x.shape
    oh, ow = pooled_size(h, k, s), pooled_size(w, k, s)
    ph, pw = (oh - 1) * s + k, (ow - 1) * s + k
    xp = np.zeros((c, ph, pw), np.float32)
    xp[:, :h, :w] = x
    valid = np.zeros((ph, pw), bool)
    valid[:h, :w] = True
    stack = np.empty((k * k, c, oh, ow), np.float32)
    vstack = np.empty((k * k, oh, ow), bool)
    for dy in range(k):
        for dx in range(k):
            stack[dy * k + dx] = xp[:, dy:dy + oh * s:s, dx:dx + ow * s:s]
            vstack[dy * k + dx] = valid[dy:dy + oh * s:s, dx:dx + ow * s:s]
    return stack, vstack


def pool_forward(x, mode='MAX', k=2, s=2):
    """Returns (y, aux).  aux = argmax window slot [C,oh,ow] for MAX, None for AVE."""
    stack, vstack = _windows(x, k, s)
    if mode == 'MAX':
        masked = np.where(vstack[:, None], stack, -np.inf).astype(np.float32)
        # np.argmax returns the FIRST maximum along the axis, which is Caffe's strict '>' scan
        # in row-major window order (dy outer, dx inner).
        arg = np.argmax(masked, axis=0)
        y = np.take_along_axis(masked, arg[None], axis=0)[0]
        return np.ascontiguousarray(y, np.float32), arg.astype(np.int8)
    if mode == 'AVE':
        cnt = vstack.sum(axis=0).astype(np.float32)
        y = (stack * vstack[:, None]).sum(axis=0, dtype=np.float32) / cnt
        return np.ascontiguousarray(y, np.float32), None
    raise ValueError(mode)


def pool_backward(dy, x_shape, aux, mode='MAX', k=2, s=2):
    """Routes dy [C,oh,ow] back to the pool input of shape x_shape."""
    c, h, w = x_shape
    oh, ow = dy.shape[-2:]
    ph, pw = (oh - 1) * s + k, (ow - 1) * s + k
    dxp = np.zeros((c, ph, pw), np.float32)
    if mode == 'MAX':
        for slot in range(k * k):
            dyy, dxx = divmod(slot, k)
            dxp[:, dyy:dyy + oh * s:s, dxx:dxx + ow * s:s] += np.where(aux == slot, dy, 0)
    else:
        valid = np.zeros((ph, pw), np.float32)
        valid[:h, :w] = 1
        cnt = np.zeros((oh, ow), np.float32)
        for dyy in range(k):
            for dxx in range(k):
                cnt += valid[dyy:dyy + oh * s:s, dxx:dxx + ow * s:s]
        share = dy / cnt
        for dyy in range(k):
            for dxx in range(k):
                dxp[:, dyy:dyy + oh * s:s, dxx:dxx + ow * s:s] += share
    return np.ascontiguousarray(dxp[:, :h, :w])
