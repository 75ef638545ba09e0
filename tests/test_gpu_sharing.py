"""Host machinery of the farm on the GPU: per-GPU sharing of weights / packed banks / targets
between the engines of a farm, the event-ordered scatter / gather of TileFarm.eval_sc_grad, the
zero-copy tile hand-over.  None of it may change a bit of the results: every check here is `==`
against the plain one-engine path."""

import numpy as np
import pytest

from tests.gpu_helpers import builtin_net, require_gpu, synthetic_weights

pytestmark = pytest.mark.gpu

CL, SL = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
CW, SW = {'conv4_2': 0.05}, {l: 0.2 for l in SL}


def _targets(eng, full_hw, rng):
    h, w = full_hw
    contents = [{l: np.abs(rng.standard_normal(eng.feature_shape(l, h, w))).astype(np.float32)
                 for l in CL}]
    styles = [{l: np.tril(rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32)
               for l in SL}]
    return contents, styles


def test_engines_of_one_gpu_share_weights_banks_and_targets():
    """A farm's engines on one GPU hold ONE copy of the weights, the packed filter banks and the
    targets, and the targets are uploaded once per GPU and scale (the reference sends them to
    every worker process: style_transfer.py:309-332)."""
    from style_transfer_amd import lib
    from style_transfer_amd.engine import TileEngine
    from style_transfer_amd.farm import TileFarm
    require_gpu()
    net = builtin_net('vgg19')
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(21)
    img = rng.uniform(-110, 120, (3, 128, 128)).astype(np.float32)

    def one_step(farm):
        eng = farm.master
        np.random.seed(3)
        contents = [farm.prepare_features_device(img, CL, 64, passes=2)]
        feats = farm.prepare_features_device(img[:, :64, :72], SL, 64, passes=1)
        farm.set_contents_and_styles(contents, [{l: farm.gram_matrix(f) for l, f in feats.items()}])
        d_img, d_grad = eng.to_device(img), eng.empty(img.shape).zero()
        loss = farm.eval_sc_grad(d_img, d_grad, (16, -8), CL, SL, {}, CW, SW, 64)     # 2 x 2 tiles
        return loss, d_grad.get()

    solo = TileFarm(net, [0], weights, verbose=False, streams_per_device=1)
    ref = one_step(solo)
    solo_bytes = solo.master.query(lib.Q_WEIGHT_BYTES)
    assert solo.master.query(lib.Q_SHARED_ENGINES) == 1
    solo.close()

    farm = TileFarm(net, [0], weights, verbose=False, streams_per_device=4)
    got = one_step(farm)
    assert len(farm.engines) == 4 and len(farm.primaries()) == 1
    for eng in farm.engines:
        assert eng.query(lib.Q_SHARED_ENGINES) == 4
        assert eng.query(lib.Q_TARGET_UPLOADS) == 1          # one upload for four engines
        assert eng.query(lib.Q_WEIGHT_BYTES) == solo_bytes   # one bank, not four
    one_step(farm)
    assert farm.master.query(lib.Q_TARGET_UPLOADS) == 2
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    # a stand-alone engine is its own group
    other = TileEngine(net, 0, weights)
    assert other.query(lib.Q_SHARED_ENGINES) == 1 and other.query(lib.Q_TARGET_UPLOADS) == 0
    other.close()
    farm.close()


def test_lazy_loss_and_stream_ordered_gradient():
    """eval_sc_grad(lazy=True) returns without a host wait; the gradient is complete in stream
    order on the master and float(loss) equals the synchronous call."""
    from style_transfer_amd.farm import LazyLoss, TileFarm
    require_gpu()
    net = builtin_net('vgg19')
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(4)
    img = rng.uniform(-110, 120, (3, 96, 160)).astype(np.float32)
    farm = TileFarm(net, [0], weights, verbose=False, force_staging=True)
    eng = farm.master
    contents = [farm.prepare_features_device(img, CL, 64, passes=1)]
    feats = farm.prepare_features_device(img[:, :64, :64], SL, 64, passes=1)
    farm.set_contents_and_styles(contents, [{l: farm.gram_matrix(f) for l, f in feats.items()}])
    d_img, g_sync, g_lazy = eng.to_device(img), eng.empty(img.shape).zero(), eng.empty(img.shape).zero()
    want = farm.eval_sc_grad(d_img, g_sync, (8, 24), CL, SL, {}, CW, SW, 64)
    for _ in range(3):
        lazy = farm.eval_sc_grad(d_img, g_lazy, (8, 24), CL, SL, {}, CW, SW, 64, lazy=True)
        assert isinstance(lazy, LazyLoss)
        copy = eng.empty(img.shape).copy_from(g_lazy)       # ordered behind the stitch on the master
        assert float(lazy) == want
        assert np.array_equal(copy.get(), g_sync.get())
        copy.free()
    farm.close()


def test_zero_copy_tiles_equal_copied_tiles():
    """One tile per engine: the master cuts straight into the engines' input blobs and stitches out
    of their gradient blobs (stx_tile_buffers).  Same bits as with separate tile buffers and the
    two device-to-device copies per tile."""
    from style_transfer_amd.farm import TileFarm
    require_gpu()
    net = builtin_net('vgg19')
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(8)
    img = rng.uniform(-110, 120, (3, 128, 144)).astype(np.float32)
    results = []
    for zero_copy in (True, False):
        farm = TileFarm(net, [0], weights, verbose=False, streams_per_device=4)     # one tile per engine
        farm.zero_copy = zero_copy
        eng = farm.master
        np.random.seed(1)
        contents = [farm.prepare_features_device(img, CL, 96, passes=2)]
        feats = farm.prepare_features_device(img[:, :64, :64], SL, 96, passes=1)
        farm.set_contents_and_styles(contents, [{l: farm.gram_matrix(f) for l, f in feats.items()}])
        d_img, d_grad = eng.to_device(img), eng.empty(img.shape).zero()
        losses = [farm.eval_sc_grad(d_img, d_grad, (8 * k, -16), CL, SL, {}, CW, SW, 96) for k in range(3)]
        assert farm.tile_evals == 12 and bool(farm._tiles) != zero_copy
        results.append((losses, d_grad.get()))
        farm.close()
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])
