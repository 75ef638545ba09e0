"""Helpers for the -m gpu parity tests (HIP path through the C ABI vs the numpy oracle)."""

import os

import numpy as np
import pytest

from oracle import layers as L
from oracle.caffe_net import synthetic_weights
from style_transfer_amd.netspec import builtin_net

_ENGINES = {}
TIGHT = 1e-5


def require_gpu():
    """-m gpu tests need a device.  On a machine that has no AMD GPU driver node (/dev/kfd) they
    are skipped; where the node exists (the GPU box) or STX_REQUIRE_GPU=1 is set, a missing
    device or extension is a FAILURE, so a silent skip cannot hide a broken HIP path there."""
    from style_transfer_amd import lib
    if lib.device_count() >= 1:
        return
    msg = 'no GPU visible: -m gpu tests need an MI355X and the built libstx.so'
    if os.environ.get('STX_REQUIRE_GPU') == '1' or os.path.exists('/dev/kfd'):
        pytest.fail(msg)
    pytest.skip(msg)


def gpu_engine(model='vgg19', seed=0):
    """A cached TileEngine with the same seeded synthetic weights the oracle uses."""
    from style_transfer_amd.engine import TileEngine
    require_gpu()
    key = (model, seed)
    if key not in _ENGINES:
        net = builtin_net(model)
        _ENGINES[key] = TileEngine(net, 0, synthetic_weights(net.as_dicts(), seed))
    return _ENGINES[key]


def max_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def l2_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _dilate3(m):
    """3x3 binary dilation (the footprint of a 3x3 convolution's backward pass)."""
    p = np.zeros((m.shape[0] + 2, m.shape[1] + 2), bool)
    p[1:-1, 1:-1] = m
    out = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            out |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
    return out


def decision_taint(layers, acts_a, acts_b, deepest, extra_shapes):
    """Which image pixels can see a DISCRETE decision on which two forward passes disagree.

    The backward pass is a discontinuous function of the activations in two places only: the
    `> 0` mask of an in-place ReLU (style_transfer.py:606-610 via Caffe's ReLU backward) and the
    argmax routing of MAX pooling.  acts_a / acts_b are {blob: [C,h,w]} of two forward passes
    (GPU and oracle).  Every position where they decide differently is marked on its blob and
    the marks are pushed down to the image through the footprints of the backward layers (3x3
    dilation per convolution, 2x2 window per pooling layer).  Returns (taint [H,W] bool on the
    image, number of differing ReLU decisions, number of differing pooling windows).  Outside
    the taint the two backward passes make identical decisions, so gradients must agree to the
    continuous tolerance there."""
    relu_after = {l['bottom'] for l in layers if l['type'] == 'ReLU'}
    stop = max(i for i, l in enumerate(layers) if l['top'] == deepest)
    taint = {}
    n_relu = n_pool = 0

    def mark(blob, m):
        taint[blob] = m if blob not in taint else (taint[blob] | m)

    def relu_flips(blob):
        return np.any((acts_a[blob] > 0) != (acts_b[blob] > 0), axis=0)

    h, w = acts_a[deepest].shape[-2:]
    mark(deepest, np.zeros((h, w), bool))
    for lay in reversed(layers[1:stop + 1]):
        t = lay['type']
        if t == 'ReLU':
            continue
        top, bottom = lay['top'], lay['bottom']
        if top not in taint:
            continue
        m = taint[top]
        if top in relu_after and top != deepest:
            f = relu_flips(top)
            n_relu += int(np.count_nonzero((acts_a[top] > 0) != (acts_b[top] > 0)))
            m = m | f
        bh, bw = (acts_a[bottom].shape if bottom in acts_a else extra_shapes[bottom])[-2:]
        if t == 'Convolution':
            mark(bottom, _dilate3(m))
        elif t == 'Pooling':
            if lay['pool'] == 'MAX':
                _, arg_a = L.pool_forward(acts_a[bottom], 'MAX')
                _, arg_b = L.pool_forward(acts_b[bottom], 'MAX')
                diff = arg_a != arg_b
                n_pool += int(np.count_nonzero(diff))
                m = m | np.any(diff, axis=0)
            up = np.repeat(np.repeat(m, 2, axis=0), 2, axis=1)[:bh, :bw]
            mark(bottom, up)
    return taint['data'], n_relu, n_pool


def loss_from_activations(om, acts, start, cl, sl, lw, cw, sw, dtype=np.float64):
    """The loss of eval_sc_grad_tile (style_transfer.py:575-593) from given activations with all
    reductions carried in `dtype`.  float64 gives the value the reference's float32 BLAS sums
    (sdot over up to 2^23 elements, ssyrk over 2^20 pixels) approximate; returns (total,
    {term: value})."""
    start = np.asarray(start)
    total, terms = 0.0, {}
    for b in om.deep_to_shallow(list(cl) + list(sl)):
        w = lw.get(b, 1.0)
        feat = np.asarray(acts[b], dtype)
        fy, fx = start // om.scale[b]
        fh, fw = feat.shape[-2:]
        if b in cl:
            for content in om.contents:
                d = (feat - content[b][:, fy:fy + fh, fx:fx + fw].astype(dtype)).ravel()
                terms['c:' + b] = w * cw[b] * float(np.dot(d, d)) / 2
                total += terms['c:' + b]
        if b in sl:
            for style in om.styles:
                f = feat.reshape(feat.shape[0], -1)
                g = np.tril(f @ f.T * dtype(1 / f.size)) - np.tril(style[b]).astype(dtype)
                v = w * sw[b] * float(np.dot(g.ravel(), g.ravel())) / 2 / len(om.styles)
                terms['s:' + b] = terms.get('s:' + b, 0.0) + v
                total += v
    return total, terms


FP32_KERNELS = {'STX_CONV_H2': '0', 'STX_GRAM': 'fp32', 'STX_SYMM': 'fp32'}


class fp32_kernels:
    """Within the block the library takes its fp32-MFMA kernels only (a new snapshot of the switches on the
    way in and out: stx_reread_env): no fp16-split convolution, Gram or SYMM -- round 4's arithmetic."""

    def __enter__(self):
        from style_transfer_amd import lib
        self.old = {k: os.environ.get(k) for k in FP32_KERNELS}
        os.environ.update(FP32_KERNELS)
        lib.reread_env()

    def __exit__(self, *exc):
        from style_transfer_amd import lib
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.reread_env()


def check_tile(eng, om, tile, start, roll, cl, cw, sl, sw, lw, ref_grad=None, flip_l2=1e-2,
               blas_loss_tol=TIGHT, fp32_leg=False):
    """GPU tile evaluation (stx_sc_grad_tile) vs the oracle.  Stated tolerances:
      * every activation: 1e-5 of max|ref|;
      * the loss: 1e-5 relative to the float64 value of the reference's formula on the oracle's
        activations, and `blas_loss_tol` relative to the oracle's own float32 result.  The
        reference sums 1/2|F - Fc|^2 with a float32 BLAS sdot (num_utils.py:69-71); over the 2^23
        elements of conv4_2 in a 1024 x 1024 tile that sum is itself only good to 0.4e-4 .. 1.2e-4
        (measured, tools/diag_loss.py: float32 2.783972e9, float64 2.784095e9, GPU 2.784095e9;
        VGG-16: float32 6.742568e8, float64 6.743383e8, GPU 6.743383e8), so the full-size cases
        pass 5e-4 there; every small case keeps 1e-5;
      * gradient vs the oracle's backward pass run on the GPU's activations (identical discrete
        decisions): 1e-5 of max|ref| per pixel;
      * gradient vs the oracle's own end-to-end result (`ref_grad` given: vs the reference's
        untouched vectors): 1e-5 of max|ref| on every pixel that cannot see a differing ReLU /
        argmax decision (decision_taint), and relative L2 < flip_l2 overall.
    fp32_leg: the forward pass is evaluated a second time with the fp32-MFMA kernels only and its decision
    flips against the same oracle pass are counted too (stats['fp32_relu_flips'], ['fp32_pool_flips'],
    ['fp32_act_err'] / ['act_err']: the largest activation error of each leg) -- the A/B behind "fp32-class".
    Returns (loss, grad, stats)."""
    loss, grad = eng.sc_grad_tile(tile, start, roll, cl, sl, lw, cw, sw)
    deepest = om.deep_to_shallow(list(cl) + list(sl))[0]
    blobs = om.blob_names[:om.blob_names.index(deepest) + 1]
    acts = eng.features_tile(tile, blobs)
    om.roll_contents(roll)
    try:
        ref_loss, oracle_grad = om.sc_grad_tile(tile, start, cl, sl, lw, cw, sw)
        ref_acts = {b: om.net.blobs[b].data[0].copy() for b in blobs}
        same_loss, same_grad = om.sc_grad_tile(tile, start, cl, sl, lw, cw, sw, activations=acts)
        loss64, _ = loss_from_activations(om, ref_acts, start, cl, sl, lw, cw, sw, np.float64)
    finally:
        om.roll_contents(-np.asarray(roll))
    act_err = 0.0
    for b in blobs:
        err = max_rel(acts[b], ref_acts[b])
        assert err < TIGHT, b
        act_err = max(act_err, err)
    assert loss == pytest.approx(loss64, rel=TIGHT), (loss, loss64, ref_loss)
    assert loss == pytest.approx(ref_loss, rel=blas_loss_tol), (loss, loss64, ref_loss)
    assert loss == pytest.approx(same_loss, rel=blas_loss_tol), (loss, loss64, same_loss)
    assert max_rel(grad, same_grad) < TIGHT
    target = oracle_grad if ref_grad is None else ref_grad
    shape = {'data': np.asarray(tile).shape}
    taint, n_relu, n_pool = decision_taint(om.net.layers, acts, ref_acts, deepest, shape)
    clean = ~taint
    stats = dict(relu_flips=n_relu, pool_flips=n_pool, tainted=float(taint.mean()),
                 l2=l2_rel(grad, target), act_err=act_err)
    if fp32_leg:
        with fp32_kernels():
            acts32 = eng.features_tile(tile, blobs)
        _, n_relu32, n_pool32 = decision_taint(om.net.layers, acts32, ref_acts, deepest, shape)
        stats.update(fp32_relu_flips=n_relu32, fp32_pool_flips=n_pool32,
                     fp32_act_err=max(max_rel(acts32[b], ref_acts[b]) for b in blobs))
        del acts32
    scale = np.abs(target).max()
    if clean.any():
        err = np.abs(np.float64(grad) - target)[:, clean].max() / scale
        stats['clean_err'] = float(err)
        assert err < TIGHT, stats
    if n_relu == 0 and n_pool == 0:
        assert max_rel(grad, target) < TIGHT
    assert stats['l2'] < flip_l2, stats
    return loss, grad, stats
