"""The per-tile hot path (stx_features_tile / stx_sc_grad_tile through the C ABI) against the
golden vectors produced by the reference's own Python and against the numpy oracle.

Stated tolerances (float32 on both sides; only summation order differs):
  * activations of every blob and the loss:      |a - ref| <= 1e-5 * max|ref|
  * gradient, AVE-pool nets (continuous):         per pixel |g - ref| <= 1e-5 * max|ref|
  * gradient, MAX-pool nets: the gradient is a DISCONTINUOUS function of the activations (argmax
    routing of max pooling, `> 0` ReLU masks).  Two float32 forward passes that agree to 1e-6
    still pick different winners in windows whose two largest values differ by less than that
    (a few windows per tile on these fixtures), and each flipped window moves O(1e-3) of the
    gradient norm.  So the check is two-sided: (a) with the oracle's backward pass given the
    GPU's activations (identical discrete decisions) the gradient must agree per pixel to
    1e-5 * max|ref|; (b) against the untouched reference vectors the relative L2 error must stay
    below 1e-2, i.e. only a handful of windows may have flipped; and (c) the differing decisions
    are located (tests/gpu_helpers.decision_taint) and every image pixel that cannot see one must
    agree with the untouched reference vectors to 1e-5 * max|ref| -- so a wrong border column or
    a mis-addressed tile cannot hide under the L2 bound."""

import numpy as np
import pytest

from tests.gpu_helpers import check_tile, gpu_engine, l2_rel, max_rel
from tests.helpers import (DEFAULT_STYLE_LAYERS, make_oracle, normalized_weights, u8_to_params)

pytestmark = pytest.mark.gpu
TIGHT = 1e-5
FLIP_L2 = 1e-2


def _targets(golden, tag, model):
    g = {k[len('tile.%s.' % tag):]: v for k, v in golden.items() if k.startswith('tile.%s.' % tag)}
    om, _ = make_oracle(model)
    tile_size = int(g['tile_size'])
    content_layers, content_weight = normalized_weights(['conv4_2'], 0.05)
    style_layers, style_weight = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    styles = [u8_to_params(g[k]) for k in sorted(g) if k.startswith('style') and k.endswith('_u8')]
    np.random.seed(123)
    om.styles = [om.style_grams(styles, style_layers, tile_size)]
    om.contents = [om.prepare_features(u8_to_params(g['content_u8']), content_layers, tile_size)]
    return g, om, (content_layers, content_weight, style_layers, style_weight)


@pytest.mark.parametrize('tag,model', [('vgg19', 'vgg19'), ('vgg16avg', 'vgg16_avgpool')])
def test_sc_grad_tile_matches_reference_vectors(golden, tag, model):
    g, om, (cl, cw, sl, sw) = _targets(golden, tag, model)
    eng = gpu_engine(model)
    eng.set_contents_and_styles(om.contents, om.styles)
    lw = {'conv3_1': float(g['lw_conv3_1'])}
    tile = np.ascontiguousarray(g['img_rolled'][:, 8:48, 16:72])
    continuous = 'avg' in model
    # the golden single tile was evaluated with the worker's content maps un-rolled (roll 0)
    loss, grad, stats = check_tile(eng, om, tile, (8, 16), (0, 0), cl, cw, sl, sw, lw,
                                   ref_grad=g['single.grad'],
                                   flip_l2=1e-3 if continuous else FLIP_L2)
    print(tag, stats)
    assert loss == pytest.approx(float(g['single.loss']), rel=TIGHT)
    if continuous:
        assert max_rel(grad, g['single.grad']) < TIGHT
    feats = eng.features_tile(tile, ['pool1', 'conv5_1'])
    assert max_rel(feats['conv5_1'], g['single.feat_conv5_1']) < TIGHT
    assert feats['pool1'].sum(dtype=np.float64) == pytest.approx(float(g['single.feat_pool1_sum']),
                                                                 rel=TIGHT)


@pytest.mark.parametrize('tag,model', [('vgg19', 'vgg19'), ('vgg16avg', 'vgg16_avgpool')])
def test_tiled_sc_grad_with_roll_matches_reference_vectors(golden, tag, model):
    """eval_sc_grad over a 2x2 tiling with a non-zero roll (style_transfer.py:614-645,230-241)."""
    from oracle.tile_path import tile_grid
    g, om, (cl, cw, sl, sw) = _targets(golden, tag, model)
    eng = gpu_engine(model)
    eng.set_contents_and_styles(om.contents, om.styles)
    lw = {'conv3_1': float(g['lw_conv3_1'])}
    img = g['img_rolled']
    continuous = 'avg' in model
    grad = np.zeros_like(img)
    loss = 0.0
    for (y0, y1, x0, x1) in tile_grid(img.shape[-2:], int(g['tile_size'])):
        tl, tg, stats = check_tile(eng, om, np.ascontiguousarray(img[:, y0:y1, x0:x1]), (y0, x0),
                                   g['roll'], cl, cw, sl, sw, lw,
                                   ref_grad=g['grad'][:, y0:y1, x0:x1],
                                   flip_l2=1e-3 if continuous else FLIP_L2)
        print(tag, (y0, x0), stats)
        loss += tl
        grad[:, y0:y1, x0:x1] = tg
    assert loss == pytest.approx(float(g['loss']), rel=TIGHT)
    if continuous:
        assert max_rel(grad, g['grad']) < TIGHT
    else:
        assert l2_rel(grad, g['grad']) < FLIP_L2


@pytest.mark.parametrize('model', ['vgg19', 'vgg16_avgpool'])
@pytest.mark.parametrize('th,tw', [(64, 80), (37, 53), (96, 96), (33, 130), (9, 13), (16, 1), (50, 44)])
def test_sc_grad_tile_odd_sizes_against_oracle(model, th, tw):
    om, _ = make_oracle(model)
    eng = gpu_engine(model)
    rng = np.random.RandomState(th)
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    full = rng.uniform(-110, 120, (3, th + 24, tw + 40)).astype(np.float32)
    style = rng.uniform(-110, 120, (3, 50, 60)).astype(np.float32)
    om.styles = [om.style_grams([style], sl, 512)]
    om.contents = [om.prepare_features(full, cl, 512)]
    eng.set_contents_and_styles(om.contents, om.styles)
    tile = np.ascontiguousarray(full[:, 16:16 + th, 8:8 + tw])
    # tiny tiles (deep blobs of 1x1 .. 2x2) have few elements per layer, so one flipped ReLU
    # decision moves more of the gradient norm: only the same-activations check is tight there
    check_tile(eng, om, tile, (16, 8), (-16, 24), cl, cw, sl, sw, {},
               flip_l2=1e-3 if 'avg' in model and th * tw > 400 else FLIP_L2)


@pytest.mark.parametrize('model', ['vgg19_big', 'vgg16_big'])
@pytest.mark.parametrize('th,tw', [(45, 61), (66, 50)])
def test_big_nets_whose_second_stage_skips_the_first_pooling(model, th, tw):
    """The reference's *_big prototxts (vgg19_big.prototxt:62) feed conv2_1 from conv1_2: pool1 is a
    dead end, every plane from conv2_x down is twice as wide and high as in the plain net, the
    content map is a quarter of the image per side.  Tile path (activations of every blob, loss,
    gradient) and feature maps -- including the dead-end pool1 -- against the oracle."""
    om, _ = make_oracle(model)
    eng = gpu_engine(model)
    rng = np.random.RandomState(th)
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    full = rng.uniform(-110, 120, (3, th + 24, tw + 40)).astype(np.float32)
    style = rng.uniform(-110, 120, (3, 50, 60)).astype(np.float32)
    om.styles = [om.style_grams([style], sl, 512)]
    om.contents = [om.prepare_features(full, cl, 512)]
    assert om.contents[0]['conv4_2'].shape[1:] == (-(-(th + 24) // 4), -(-(tw + 40) // 4))
    eng.set_contents_and_styles(om.contents, om.styles)
    tile = np.ascontiguousarray(full[:, 16:16 + th, 8:8 + tw])
    check_tile(eng, om, tile, (16, 8), (-16, 24), cl, cw, sl, sw, {})
    wanted = ['pool1', 'conv2_1', 'conv5_1']
    feats, ref = eng.features_tile(tile, wanted), om.features_tile(tile, wanted)
    assert feats['conv2_1'].shape[1:] == (th, tw) and feats['pool1'].shape[1:] == (-(-th // 2), -(-tw // 2))
    for l in wanted:
        assert max_rel(feats[l], ref[l]) < TIGHT, l


def test_non_default_taps(golden):
    """Content on a pooling blob, style on conv1_2/conv3_3 only, weighted layers."""
    om, _ = make_oracle('vgg16_avgpool')
    eng = gpu_engine('vgg16_avgpool')
    rng = np.random.RandomState(3)
    cl, cw = ['pool3'], {'pool3': 0.3}
    sl, sw = ['conv1_2', 'conv3_3'], {'conv1_2': 0.25, 'conv3_3': 0.75}
    full = rng.uniform(-110, 120, (3, 72, 88)).astype(np.float32)
    style = rng.uniform(-110, 120, (3, 40, 44)).astype(np.float32)
    om.styles = [om.style_grams([style], sl, 512)]
    om.contents = [om.prepare_features(full, cl, 512)]
    eng.set_contents_and_styles(om.contents, om.styles)
    tile = np.ascontiguousarray(full[:, 8:8 + 56, 16:16 + 64])
    check_tile(eng, om, tile, (8, 16), (0, 0), cl, cw, sl, sw, {'conv1_2': 2.0, 'pool3': 0.5},
               flip_l2=1e-3)


def test_errors_are_reported_not_fatal():
    from style_transfer_amd.lib import StxError
    eng = gpu_engine('vgg19')
    tile = np.zeros((3, 32, 32), np.float32)
    with pytest.raises(StxError):
        eng.features_tile(tile, ['no_such_layer'])
    eng.set_contents_and_styles([], [])
    with pytest.raises(StxError):
        eng.sc_grad_tile(tile, (0, 0), (0, 0), ['conv4_2'], [], {}, {'conv4_2': 1.0}, {})


@pytest.mark.parametrize('tag,model', [('vgg19', 'vgg19'), ('vgg16avg', 'vgg16_avgpool')])
def test_deep_dream_term_matches_reference_vectors(golden, tag, model):
    """dd_layers / dd_weight (style_transfer.py:602-604): loss -= lw*dd*1/2|F|^2 and
    diff -= lw*dd*normalize(F) on two layers, against the reference's own eval_sc_grad_tile."""
    g, om, (cl, cw, sl, sw) = _targets(golden, tag, model)
    eng = gpu_engine(model)
    eng.set_contents_and_styles(om.contents, om.styles)
    lw = {'conv3_1': float(g['lw_conv3_1'])}
    dl, dw = ['conv4_3', 'conv2_2'], {'conv4_3': 0.01, 'conv2_2': 0.03}     # parse_weights(.., 0.04)
    tile = np.ascontiguousarray(g['img_rolled'][:, 8:48, 16:72])
    loss, grad = eng.sc_grad_tile(tile, (8, 16), (0, 0), cl, sl, lw, cw, sw, dd_layers=dl,
                                  dd_weight=dw)
    assert loss == pytest.approx(float(g['dream.loss']), rel=TIGHT)
    ref_loss, ref_grad = om.sc_grad_tile(tile, (8, 16), cl, sl, lw, cw, sw, dd_layers=dl, dd_weight=dw)
    assert ref_loss == pytest.approx(float(g['dream.loss']), rel=TIGHT)        # oracle == reference
    assert max_rel(ref_grad, g['dream.grad']) < TIGHT
    if 'avg' in model:
        assert max_rel(grad, g['dream.grad']) < TIGHT
    else:
        assert l2_rel(grad, g['dream.grad']) < FLIP_L2
    # the term really is there: without it the loss is the plain fixture's
    plain, _ = eng.sc_grad_tile(tile, (8, 16), (0, 0), cl, sl, lw, cw, sw)
    assert plain == pytest.approx(float(g['single.loss']), rel=TIGHT) and plain != loss


@pytest.mark.parametrize('model,th,tw', [('vgg19', 96, 200), ('vgg16_avgpool', 131, 77), ('vgg19', 256, 384)])
def test_first_layer_kernel_with_its_own_gram_partials(model, th, tw, monkeypatch):
    """conv_first.hip computes conv1_1 and, when the blob is a style tap, the Gram partials of its
    own output (one partial tile per workgroup, finished by the usual gram_finish).  Against the
    separate path -- conv_mfma's first-layer configuration + gram_partial_bf3 over the blob
    (STX_CONV_FIRST_FUSED=0): the blob is BIT-IDENTICAL (the same MFMA sequence: input planes in
    pairs, tap by tap), so no ReLU / pooling decision can move; the Gram differs in summation
    order only (fp32-class both ways): loss to 1e-6, gradient to 1e-5 of its maximum.  (Both are
    held to the oracle by every other test of this file.)"""
    from style_transfer_amd.engine import TileEngine
    from tests.gpu_helpers import builtin_net, require_gpu, synthetic_weights
    require_gpu()
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(th)
    tile = rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)
    cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1']
    cw, sw = {'conv4_2': 0.05}, {l: 1 / 3 for l in sl}
    out = {}
    for fused in ('0', '1'):
        monkeypatch.setenv('STX_CONV_FIRST_FUSED', fused)
        eng = TileEngine(net, 0, weights)
        r = np.random.RandomState(3)
        eng.set_contents_and_styles(
            [{l: np.abs(r.standard_normal(eng.feature_shape(l, th, tw))).astype(np.float32) for l in cl}],
            [{l: np.tril(r.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}
             for _ in range(2)])            # two style targets: the partials are finished twice
        loss, grad = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
        again = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
        assert again[0] == loss and np.array_equal(again[1], grad)        # deterministic
        out[fused] = (loss, grad, eng.features_tile(tile, ['conv1_1'])['conv1_1'])
        eng.close()
    assert np.array_equal(out['0'][2], out['1'][2])
    assert out['1'][0] == pytest.approx(out['0'][0], rel=1e-6)
    assert np.abs(out['0'][1] - out['1'][1]).max() <= 1e-5 * np.abs(out['0'][1]).max()


@pytest.mark.parametrize('model,th,tw', [('vgg19', 256, 384), ('vgg19', 203, 331), ('vgg16_avgpool', 256, 320),
                                         ('vgg16_avgpool', 131, 277), ('vgg19', 64, 96)])
def test_pooling_backward_inside_the_next_convolution(model, th, tw, monkeypatch):
    """The backward pass of a convolution that sits under a 2x2/2 pooling layer takes the POOLED gradient
    and the window codes and routes it inside its own patch staging (conv_h2.hip, PIN): the pooling
    layer's backward kernel does not run and the four-times-larger gradient of the convolution's output
    is never written.  Same numbers into the same arithmetic: against STX_POOL_BWD_FUSE=0 (the stand-alone
    kernel, pool.hip -- held to the oracle by every other test here) the gradient is BIT-IDENTICAL, on
    MAX and AVE nets, even and odd planes (windows cut by the right / bottom border)."""
    from style_transfer_amd.engine import TileEngine
    from tests.gpu_helpers import builtin_net, require_gpu, synthetic_weights
    require_gpu()
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(th)
    tile = rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)
    cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
    cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
    out, skipped = {}, {}
    for fuse in ('0', '1'):
        monkeypatch.setenv('STX_POOL_BWD_FUSE', fuse)
        eng = TileEngine(net, 0, weights)
        r = np.random.RandomState(3)
        eng.set_contents_and_styles(
            [{l: np.abs(r.standard_normal(eng.feature_shape(l, th, tw))).astype(np.float32) for l in cl}],
            [{l: np.tril(r.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}])
        eng.profile(True)
        out[fuse] = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
        labels = [row[0] for row in eng.profile_read()]
        skipped[fuse] = 4 - sum(l.startswith('bwd pool') for l in labels)
        eng.profile(False)
        again = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
        assert again[0] == out[fuse][0] and np.array_equal(again[1], out[fuse][1])
        eng.close()
    assert skipped['0'] == 0
    # (the small planes of the deep layers split their reduction and keep the stand-alone kernel)
    assert skipped['1'] >= (1 if th * tw < 10000 else 2), skipped
    assert out['1'][0] == out['0'][0]
    assert np.array_equal(out['1'][1], out['0'][1])


@pytest.mark.parametrize('model,th,tw', [('vgg19', 203, 331), ('vgg16_avgpool', 131, 277), ('vgg19', 256, 384),
                                         ('vgg16_avgpool', 90, 61)])
def test_pooling_forward_inside_the_convolution_epilogue(model, th, tw, monkeypatch):
    """The 2x2/2 pooling layer behind a convolution runs in that convolution's epilogue (a lane's 2 x 2
    outputs are one window) -- since round 5 on odd planes too, where the last window of a row has one
    column as the last of a column may have one row (ceil mode: MAX over what exists, AVE with the clipped
    window's divisor).  pool.hip's arithmetic on the same values: against STX_POOL_FWD_FUSE=0 (the
    stand-alone kernel) loss, gradient and every pooled blob are BIT-IDENTICAL."""
    from style_transfer_amd.engine import TileEngine
    from tests.gpu_helpers import builtin_net, require_gpu, synthetic_weights
    require_gpu()
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(tw)
    tile = rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)
    cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
    cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
    pools = ['pool1', 'pool2', 'pool3', 'pool4']
    out, standalone = {}, {}
    for fuse in ('0', '1'):
        monkeypatch.setenv('STX_POOL_FWD_FUSE', fuse)
        eng = TileEngine(net, 0, weights)
        r = np.random.RandomState(3)
        eng.set_contents_and_styles(
            [{l: np.abs(r.standard_normal(eng.feature_shape(l, th, tw))).astype(np.float32) for l in cl}],
            [{l: np.tril(r.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}])
        eng.profile(True)
        loss, grad = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
        standalone[fuse] = sum(row[0].startswith('fwd pool') for row in eng.profile_read())
        eng.profile(False)
        out[fuse] = (loss, grad, eng.features_tile(tile, pools))
        eng.close()
    assert standalone['0'] == 4
    # (planes whose convolution splits its reduction keep the stand-alone kernel)
    assert standalone['1'] <= (2 if th * tw > 10000 else 3), standalone
    assert out['1'][0] == out['0'][0] and np.array_equal(out['1'][1], out['0'][1])
    for name in pools:
        assert np.array_equal(out['1'][2][name], out['0'][2][name]), name


@pytest.mark.parametrize('model,th,tw', [('vgg19', 256, 256), ('vgg16_avgpool', 128, 192), ('vgg19', 64, 96)])
def test_vectorised_k_slice_reduce_changes_no_bit(model, th, tw, monkeypatch):
    """The pass that adds the K slices of a split convolution up (bias + ReLU, or mask + loss terms) takes
    four elements per thread where the plane size allows: every element is still the sum of its slices in
    slice order -- against STX_REDUCE_VEC=0 (one element per thread) loss and gradient are BIT-IDENTICAL,
    content- and style-injecting layers included (small planes: most layers of these tiles split)."""
    from style_transfer_amd.engine import TileEngine
    from tests.gpu_helpers import builtin_net, require_gpu, synthetic_weights
    require_gpu()
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(th + tw)
    tile = rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)
    cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
    cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
    out, reduces = {}, {}
    for vec in ('0', '1'):
        monkeypatch.setenv('STX_REDUCE_VEC', vec)
        eng = TileEngine(net, 0, weights)
        r = np.random.RandomState(3)
        eng.set_contents_and_styles(
            [{l: np.abs(r.standard_normal(eng.feature_shape(l, th, tw))).astype(np.float32) for l in cl}],
            [{l: np.tril(r.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}])
        out[vec] = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
        eng.close()
    assert out['1'][0] == out['0'][0]
    assert np.array_equal(out['1'][1], out['0'][1])
