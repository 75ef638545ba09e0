"""Plane sets of 2 GiB and more (`--tile-size` above 2896: style_transfer.py:619-632 accepts any
tile size).  The kernels address memory through buffer descriptors with 32-bit offsets; their BIG
variants move the descriptor's 64-bit base instead (per chunk of input planes, per output
channel).  Two kinds of checks:

  * the BIG variants forced on ordinary planes (STX_WINO_BIG=1, read at every launch): loss and
    gradient of whole tile evaluations must be BIT-IDENTICAL to the plain variants' -- same
    arithmetic, only the address formation differs -- so every parity result of the other test
    files carries over;
  * single kernels on a 64-channel 2944 x 2944 plane set (2.2 GB, past the limit) against the
    oracle evaluated in row strips / float64: that is where an overflowing offset would show.

The whole VGG-19 evaluation of a 2944 x 2944 tile against the oracle (a quarter of an hour of
host time) runs with STX_TEST_HUGE=1."""

import os

import numpy as np
import pytest

from oracle import layers as L
from style_transfer_amd import lib
from tests.gpu_helpers import builtin_net, check_tile, gpu_engine, max_rel, require_gpu, synthetic_weights
from tests.helpers import DEFAULT_STYLE_LAYERS, make_oracle, normalized_weights

pytestmark = pytest.mark.gpu

CL, SL = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
CW, SW = {'conv4_2': 0.05}, {l: 0.2 for l in SL}


@pytest.mark.parametrize('model,th,tw', [('vgg19', 96, 128), ('vgg19', 181, 90), ('vgg16_avgpool', 130, 66),
                                         ('vgg19', 724, 724)])
def test_big_addressing_variants_are_bit_identical(model, th, tw, monkeypatch):
    from style_transfer_amd.engine import TileEngine
    require_gpu()
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(th)
    tile = rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)
    results = []
    for big in ('0', '1'):
        monkeypatch.setenv('STX_WINO_BIG', big)
        eng = TileEngine(net, 0, weights)
        r = np.random.RandomState(7)
        eng.set_contents_and_styles(
            [{l: np.abs(r.standard_normal(eng.feature_shape(l, th + 16, tw + 8))).astype(np.float32) for l in CL}],
            [{l: np.tril(r.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in SL}])
        loss, grad = eng.sc_grad_tile(tile, (8, 0), (-24, 16), CL, SL, {'conv2_1': 0.5}, CW, SW)
        feats = eng.features_tile(tile, ['conv1_2', 'pool2', 'conv3_3'])
        results.append((loss, grad, feats))
        eng.close()
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])
    for k in results[0][2]:
        assert np.array_equal(results[0][2][k], results[1][2][k]), k


def _conv_strips(x, wt, b, rows=256):
    """oracle conv_forward in row strips (same arithmetic per output; bounded memory)."""
    cin, h, w = x.shape
    out = np.empty((wt.shape[0], h, w), np.float32)
    xp = np.zeros((cin, h + 2, w), np.float32)
    xp[:, 1:-1] = x
    for y0 in range(0, h, rows):
        y1 = min(h, y0 + rows)
        strip = L.conv_forward(xp[:, y0:y1 + 2], wt, b)           # pads rows again: drop them
        out[:, y0:y1] = strip[:, 1:-1]
    return out


def test_convolution_on_a_plane_set_beyond_2_gib():
    """64 -> 64 channels on 2944 x 2944 (2.2 GB in, 2.2 GB out): forward + bias + ReLU, and the
    backward pass with the ReLU mask, against the oracle in strips (2e-5, the kernel tests' bound)."""
    eng = gpu_engine()
    rng = np.random.default_rng(3)
    c, h, w = 64, 2944, 2944
    x = np.maximum(rng.standard_normal((c, h, w), dtype=np.float32), 0)
    wt = (rng.standard_normal((c, c, 3, 3)) * np.sqrt(2 / (9 * c))).astype(np.float32)
    b = (0.1 * rng.standard_normal(c)).astype(np.float32)
    dx_, dw, db = eng.to_device(x), eng.to_device(wt), eng.to_device(b)
    y = eng.empty((c, h, w))
    lib.call('stx_op_conv_forward', eng.handle, dx_.ptr, c, h, w, dw.ptr, db.ptr, c, 3, 1, y.ptr)
    got = y.get()
    ref = np.maximum(_conv_strips(x, wt, b), 0)
    assert max_rel(got, ref) < 2e-5
    # corners and the last rows / columns explicitly (the far end of the address range)
    assert max_rel(got[-1, -8:, -64:], ref[-1, -8:, -64:]) < 2e-5
    del got, ref
    dy = rng.standard_normal((c, h, w), dtype=np.float32)
    ddy = y.set(dy)
    gx = eng.empty((c, h, w))
    lib.call('stx_op_conv_backward_data', eng.handle, ddy.ptr, c, h, w, dw.ptr, c, 3, dx_.ptr, gx.ptr)
    wt_b = np.ascontiguousarray(wt.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1])      # dgrad = conv with W^T rot180
    ref = _conv_strips(dy, wt_b, np.zeros(c, np.float32)) * (x > 0)
    assert max_rel(gx.get(), ref) < 2e-5
    for a in (dx_, dw, db, y, gx):
        a.free()


def test_style_terms_and_last_layer_on_a_plane_set_beyond_2_gib():
    """conv1_1's loss terms (Gram, G - Gs, S = sym(D) F, sum |S|) and the backward pass into the
    image (64 -> 3) on 64 x 2944 x 2944."""
    import ctypes
    eng = gpu_engine()
    rng = np.random.default_rng(4)
    c, h, w = 64, 2944, 2944
    feat = np.maximum(rng.standard_normal((c, h, w), dtype=np.float32) * 20 + 4, 0)
    target = np.tril(rng.standard_normal((c, c))).astype(np.float32)
    d_feat, d_tgt, d_s = eng.to_device(feat), eng.to_device(target), eng.empty((c, h, w))
    half_sq, abs_sum = ctypes.c_double(0), ctypes.c_double(0)
    lib.call('stx_op_style_terms', eng.handle, d_feat.ptr, c, h, w, d_tgt.ptr, d_s.ptr, None,
             ctypes.byref(half_sq), ctypes.byref(abs_sum))
    f64 = feat.reshape(c, -1).astype(np.float64)
    d = np.tril(f64 @ f64.T / feat.size) - target
    assert half_sq.value == pytest.approx(0.5 * float((d * d).sum()), rel=2e-5)
    dsym = d + np.tril(d, -1).T
    s_ref = dsym @ f64
    got = d_s.get().reshape(c, -1)
    assert max_rel(got, s_ref) < 2e-5
    assert abs_sum.value == pytest.approx(float(np.abs(s_ref).sum()), rel=2e-5)
    del got, s_ref, f64
    # backward into the image: the 4x4x1-MFMA kernel reads all 64 planes
    wt = (rng.standard_normal((64, 3, 3, 3)) * np.sqrt(2 / 27)).astype(np.float32)
    dw, gx = eng.to_device(wt), eng.empty((3, h, w))
    lib.call('stx_op_conv_backward_data', eng.handle, d_feat.ptr, 64, h, w, dw.ptr, 3, 3, None, gx.ptr)
    wt_b = np.ascontiguousarray(wt.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1])
    ref = _conv_strips(feat, wt_b, np.zeros(3, np.float32))
    assert max_rel(gx.get(), ref) < 2e-5
    for a in (d_feat, d_tgt, d_s, dw, gx):
        a.free()


@pytest.mark.skipif(os.environ.get('STX_TEST_HUGE') != '1', reason='a quarter of an hour of oracle time: STX_TEST_HUGE=1')
def test_sc_grad_tile_2944():
    """`--size 2944 --tile-size 2944`: one seam-free tile whose conv1_x blobs are 2.2 GB each."""
    om, _ = make_oracle('vgg19')
    eng = gpu_engine('vgg19')
    rng = np.random.RandomState(11)
    th = tw = 2944
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    contents = {l: np.abs(rng.standard_normal((om.channels[l], -(-th // om.scale[l]), -(-tw // om.scale[l])))).astype(np.float32)
                for l in cl}
    styles = {l: np.tril(0.05 * rng.standard_normal((om.channels[l],) * 2)).astype(np.float32) for l in sl}
    om.contents, om.styles = [contents], [styles]
    eng.set_contents_and_styles(om.contents, om.styles)
    coarse = rng.uniform(-110, 120, (3, th // 16 + 2, tw // 16 + 2)).astype(np.float32)
    tile = np.repeat(np.repeat(coarse, 16, axis=1), 16, axis=2)[:, :th, :tw]
    tile = np.ascontiguousarray(tile + rng.uniform(-16, 16, (3, th, tw)).astype(np.float32))
    _, _, stats = check_tile(eng, om, tile, (0, 0), (-1000, 344), cl, cw, sl, sw, {}, blas_loss_tol=5e-4, fp32_leg=True)
    line = '2944x2944 tile: %s, %.2f ms on the GPU' % (stats, eng.last_tile_ms())
    print(line)
    if os.environ.get('STX_PARITY_STATS'):
        with open(os.environ['STX_PARITY_STATS'], 'a') as f:
            f.write(line + '\n')
    flips, flips32 = stats['relu_flips'] + stats['pool_flips'], stats['fp32_relu_flips'] + stats['fp32_pool_flips']
    assert flips < 500 * th * tw / 2 ** 20, stats
    assert stats['tainted'] < 0.3, stats
    assert flips <= 2 * flips32 + 50 and stats['act_err'] <= 2 * stats['fp32_act_err'] + 1e-7, stats


def test_sc_grad_tile_past_2gib_shallow_taps():
    """Un-gated whole-tile case past the 32-bit addressing limit: a 2896 x 2912 tile (8.43 M
    pixels: its 64-channel conv1_x plane sets are 2.16 GB each) through stx_sc_grad_tile with the
    content tap at conv2_2 and style taps at conv1_1 / conv2_1, so that the oracle runs four
    convolution layers instead of sixteen and finishes in well under a minute.  Everything that
    has a >= 2 GiB form is on this path: conv_wino2's re-based descriptors forward and backward
    (conv1_2, mask, fused pooling), the 64-bit-indexed Gram and the BIG SYMM of conv1_1, the
    size_t pooling, the first layer and the backward pass into the image.  Same bounds as every
    other tile case (tests/gpu_helpers.check_tile).  Reference: style_transfer.py:556-612,619-632
    (any --tile-size is accepted)."""
    om, _ = make_oracle('vgg19')
    eng = gpu_engine('vgg19')
    rng = np.random.RandomState(12)
    th, tw = 2896, 2912
    assert 64 * th * tw * 4 >= 2 ** 31
    cl, cw = normalized_weights(['conv2_2'], 0.05)
    sl, sw = normalized_weights(['conv1_1', 'conv2_1'], 1)
    contents = {l: np.abs(rng.standard_normal((om.channels[l], -(-th // om.scale[l]) + 3, -(-tw // om.scale[l]) + 5))).astype(np.float32)
                for l in cl}
    styles = {l: np.tril(0.05 * rng.standard_normal((om.channels[l],) * 2)).astype(np.float32) for l in sl}
    om.contents, om.styles = [contents], [styles]
    eng.set_contents_and_styles(om.contents, om.styles)
    coarse = rng.uniform(-110, 120, (3, th // 16 + 2, tw // 16 + 2)).astype(np.float32)
    tile = np.repeat(np.repeat(coarse, 16, axis=1), 16, axis=2)[:, :th, :tw]
    tile = np.ascontiguousarray(tile + rng.uniform(-16, 16, (3, th, tw)).astype(np.float32))
    # (the loss is held to 1e-5 of the float64 value of the reference's formula; against the oracle's
    # own float32 result only to 5e-3: its BLAS sdot over the 2.7e8 elements of conv2_2 at this size
    # is off by 2.1e-3 -- measured here: GPU 2.91983468e10, float64 2.91983470e10, float32 2.9136e10)
    _, _, stats = check_tile(eng, om, tile, (4, 8), (-1000, 344), cl, cw, sl, sw, {}, blas_loss_tol=5e-3)
    print('%dx%d tile, taps up to conv2_2: %s, %.2f ms on the GPU' % (th, tw, stats, eng.last_tile_ms()))
    assert stats['relu_flips'] + stats['pool_flips'] < 500 * th * tw / 2 ** 20, stats
    assert stats['tainted'] < 0.3, stats
