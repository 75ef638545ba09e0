"""Each HIP kernel against the oracle's Caffe-layer arithmetic, through the C ABI test hooks.

Tolerance (float32): max |a - b| <= 2e-5 * max |b| for the MFMA convolutions (both sides are
fp32 FMA chains that differ only in summation order), exact for pooling."""

import ctypes

import numpy as np
import pytest

from oracle import layers as L
from oracle import num_ops
from style_transfer_amd import lib
from tests.gpu_helpers import gpu_engine, max_rel

pytestmark = pytest.mark.gpu

CONV_CASES = [  # (Cin, Cout, H, W): SURVEY section 8c list + shapes that hit every tile config
    (3, 64, 33, 47), (64, 64, 32, 32), (256, 512, 9, 11), (64, 128, 70, 65), (128, 256, 40, 40),
    (512, 512, 16, 16), (64, 3, 37, 50), (128, 64, 24, 72), (20, 36, 19, 31), (16, 70, 13, 200),
    (72, 40, 130, 129)]
# every convolution kernel family on every shape it accepts; None = the engine's own choice
CONV_ALGOS = [None, 'direct', 'wino1', 'wino2a', 'wino2b']


@pytest.mark.parametrize('algo', CONV_ALGOS)
@pytest.mark.parametrize('cin,cout,h,w', CONV_CASES)
def test_conv_forward_and_backward_data(cin, cout, h, w, algo, monkeypatch):
    if algo:
        monkeypatch.setenv('STX_CONV_ALGO', algo)     # read by the library at every call
    else:
        monkeypatch.delenv('STX_CONV_ALGO', raising=False)
    eng = gpu_engine()
    rng = np.random.RandomState(cin * 7 + cout + h)
    x = rng.standard_normal((cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2 / (9 * cin))).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    dy = rng.standard_normal((cout, h, w)).astype(np.float32)
    dx_, dw, db = eng.to_device(x), eng.to_device(wt), eng.to_device(b)
    y = eng.empty((cout, h, w))
    lib.call('stx_op_conv_forward', eng.handle, dx_.ptr, cin, h, w, dw.ptr, db.ptr, cout, 3, 1,
             y.ptr)
    ref = np.maximum(L.conv_forward(x, wt, b), 0)
    assert max_rel(y.get(), ref) < 2e-5
    lib.call('stx_op_conv_forward', eng.handle, dx_.ptr, cin, h, w, dw.ptr, db.ptr, cout, 3, 0,
             y.ptr)
    assert max_rel(y.get(), L.conv_forward(x, wt, b)) < 2e-5
    # backward to data, with and without the ReLU mask of the blob below
    ddy, gx = eng.to_device(dy), eng.empty((cin, h, w))
    lib.call('stx_op_conv_backward_data', eng.handle, ddy.ptr, cout, h, w, dw.ptr, cin, 3, None,
             gx.ptr)
    ref = L.conv_backward_data(dy, wt)
    assert max_rel(gx.get(), ref) < 2e-5
    below = np.maximum(x, 0)
    dbelow = eng.to_device(below)
    lib.call('stx_op_conv_backward_data', eng.handle, ddy.ptr, cout, h, w, dw.ptr, cin, 3,
             dbelow.ptr, gx.ptr)
    assert max_rel(gx.get(), ref * (below > 0)) < 2e-5


@pytest.mark.parametrize('h,w', [(8, 8), (7, 9), (1, 5), (33, 2), (64, 96), (543, 37)])
@pytest.mark.parametrize('mode', ['MAX', 'AVE'])
def test_pooling(h, w, mode):
    eng = gpu_engine()
    rng = np.random.RandomState(h * 3 + w)
    x = np.maximum(rng.standard_normal((6, h, w)), 0).astype(np.float32)
    x[:, ::3] = 0          # whole windows of zeros: MAX must route to the FIRST element
    code = lib.POOL_MAX if mode == 'MAX' else lib.POOL_AVE
    ref, aux = L.pool_forward(x, mode)
    dx_, y = eng.to_device(x), eng.empty(ref.shape)
    lib.call('stx_op_pool_forward', eng.handle, dx_.ptr, 6, h, w, code, y.ptr)
    assert np.array_equal(y.get(), ref) if mode == 'MAX' else max_rel(y.get(), ref) < 1e-6
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    ddy, gx = eng.to_device(dy), eng.empty(x.shape)
    lib.call('stx_op_pool_backward', eng.handle, ddy.ptr, dx_.ptr, 6, h, w, code, None, gx.ptr)
    gref = L.pool_backward(dy, x.shape, aux, mode)
    assert max_rel(gx.get(), gref) < 1e-6
    lib.call('stx_op_pool_backward', eng.handle, ddy.ptr, dx_.ptr, 6, h, w, code, dx_.ptr, gx.ptr)
    assert max_rel(gx.get(), gref * (x > 0)) < 1e-6


@pytest.mark.parametrize('c,h,w', [(64, 17, 23), (128, 40, 33), (512, 8, 9), (256, 64, 64)])
def test_gram_matrix_lower_triangle(c, h, w):
    eng = gpu_engine()
    rng = np.random.RandomState(c + h)
    feat = np.maximum(rng.standard_normal((c, h, w)), 0).astype(np.float32)
    gram = eng.gram_matrix(feat)
    assert np.all(np.triu(gram, 1) == 0)
    assert max_rel(gram, num_ops.gram_lower(feat)) < 2e-5


def test_gram_matches_reference_fixture(golden):
    eng = gpu_engine()
    assert max_rel(eng.gram_matrix(golden['num.feat']), golden['num.gram']) < 2e-5
