"""The tile path is bit-for-bit reproducible from run to run: every reduction (split-K slices,
Gram partial tiles, loss sums) is added in a fixed order, nothing uses floating-point atomics.
(The reference's Caffe path is deterministic too; tools/tail_stress.py is the long version.)"""
import hashlib

import numpy as np
import pytest

from tests.gpu_helpers import gpu_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('th,tw', [(96, 96), (37, 53), (181, 181)])
def test_tile_gradient_is_reproducible(th, tw):
    eng = gpu_engine('vgg19')
    rng = np.random.RandomState(th)
    cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
    cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
    eng.set_contents_and_styles(
        [{l: np.abs(rng.standard_normal(eng.feature_shape(l, th, tw))).astype(np.float32) for l in cl}],
        [{l: np.tril(rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}])
    tile = eng.to_device(rng.uniform(-120, 120, (3, th, tw)).astype(np.float32))
    grad = eng.empty((3, th, tw))
    seen = set()
    losses = set()
    for _ in range(25):
        p = eng.sc_grad_tile_async(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw, grad_out=grad)
        eng.sync()
        seen.add(hashlib.md5(grad.get().tobytes()).hexdigest())
        losses.add(float(p.loss))
    assert len(seen) == 1 and len(losses) == 1


@pytest.mark.parametrize('model', ['vgg19', 'vgg16_avgpool'])
@pytest.mark.parametrize('th,tw', [(96, 96), (37, 53), (130, 66)])
def test_backward_pooling_from_window_codes_equals_recomputed_argmax(model, th, tw, monkeypatch):
    """The backward pooling that runs from the one-byte window codes of the forward pass (pool.hip;
    written by the pooling kernel or by the convolution epilogue that fuses it) routes exactly the
    same gradient as the one that recomputes argmax and ReLU mask from the pool input
    (STX_POOL_CODES=0, read when an engine is created): bit-identical loss and gradient, MAX and
    AVE pooling, fused (even planes) and stand-alone (odd planes) forward pooling."""
    from tests.gpu_helpers import builtin_net, synthetic_weights
    from style_transfer_amd.engine import TileEngine
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(tw)
    cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
    cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
    results = []
    for codes in ('1', '0'):
        monkeypatch.setenv('STX_POOL_CODES', codes)
        eng = TileEngine(net, 0, weights)
        r = np.random.RandomState(7)
        eng.set_contents_and_styles(
            [{l: np.abs(r.standard_normal(eng.feature_shape(l, th, tw))).astype(np.float32) for l in cl}],
            [{l: np.tril(r.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}])
        tile = eng.to_device(np.random.RandomState(3).uniform(-120, 120, (3, th, tw)).astype(np.float32))
        grad = eng.empty((3, th, tw))
        p = eng.sc_grad_tile_async(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw, grad_out=grad)
        eng.sync()
        results.append((float(p.loss), grad.get().copy()))
        eng.close()
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])
