"""The per-tile hot path (stx_features_tile / stx_sc_grad_tile through the C ABI) against the
golden vectors produced by the reference's own Python and against the numpy oracle.

Stated tolerance: per-pixel |grad - ref| <= 1e-4 * max|ref| and loss within 1e-4 relative
(float32 end to end on both sides; only the summation order inside convolutions, Grams and
reductions differs)."""

import numpy as np
import pytest

from tests.gpu_helpers import gpu_engine, max_rel
from tests.helpers import (DEFAULT_STYLE_LAYERS, make_oracle, normalized_weights, u8_to_params)

pytestmark = pytest.mark.gpu
TOL = 1e-4


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
    # the golden single tile was evaluated with the worker's content maps un-rolled (roll 0)
    loss, grad = eng.sc_grad_tile(tile, (8, 16), (0, 0), cl, sl, lw, cw, sw)
    assert loss == pytest.approx(float(g['single.loss']), rel=TOL)
    assert max_rel(grad, g['single.grad']) < TOL
    feats = eng.features_tile(tile, ['pool1', 'conv5_1'])
    assert max_rel(feats['conv5_1'], g['single.feat_conv5_1']) < TOL
    assert feats['pool1'].sum(dtype=np.float64) == pytest.approx(float(g['single.feat_pool1_sum']),
                                                                 rel=TOL)


@pytest.mark.parametrize('tag,model', [('vgg19', 'vgg19'), ('vgg16avg', 'vgg16_avgpool')])
def test_tiled_sc_grad_with_roll_matches_reference_vectors(golden, tag, model):
    """eval_sc_grad over a 2x2 tiling with a non-zero roll (style_transfer.py:614-645,230-241)."""
    from oracle.tile_path import tile_grid
    g, om, (cl, cw, sl, sw) = _targets(golden, tag, model)
    eng = gpu_engine(model)
    eng.set_contents_and_styles(om.contents, om.styles)
    lw = {'conv3_1': float(g['lw_conv3_1'])}
    img = g['img_rolled']
    grad = np.zeros_like(img)
    loss = 0.0
    for (y0, y1, x0, x1) in tile_grid(img.shape[-2:], int(g['tile_size'])):
        tl, tg = eng.sc_grad_tile(np.ascontiguousarray(img[:, y0:y1, x0:x1]), (y0, x0), g['roll'],
                                  cl, sl, lw, cw, sw)
        loss += tl
        grad[:, y0:y1, x0:x1] = tg
    assert loss == pytest.approx(float(g['loss']), rel=TOL)
    assert max_rel(grad, g['grad']) < TOL


@pytest.mark.parametrize('th,tw', [(64, 80), (37, 53), (96, 96)])
def test_sc_grad_tile_odd_sizes_against_oracle(th, tw):
    om, _ = make_oracle('vgg19')
    eng = gpu_engine('vgg19')
    rng = np.random.RandomState(th)
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    full = rng.uniform(-110, 120, (3, th + 24, tw + 40)).astype(np.float32)
    style = rng.uniform(-110, 120, (3, 50, 60)).astype(np.float32)
    om.styles = [om.style_grams([style], sl, 512)]
    om.contents = [om.prepare_features(full, cl, 512)]
    eng.set_contents_and_styles(om.contents, om.styles)
    tile = np.ascontiguousarray(full[:, 16:16 + th, 8:8 + tw])
    roll = (-16, 24)
    om.roll_contents(roll)
    ref_loss, ref_grad = om.sc_grad_tile(tile, (16, 8), cl, sl, {}, cw, sw)
    loss, grad = eng.sc_grad_tile(tile, (16, 8), roll, cl, sl, {}, cw, sw)
    assert loss == pytest.approx(ref_loss, rel=TOL)
    assert max_rel(grad, ref_grad) < TOL


def test_errors_are_reported_not_fatal():
    from style_transfer_amd.lib import StxError
    eng = gpu_engine('vgg19')
    tile = np.zeros((3, 32, 32), np.float32)
    with pytest.raises(StxError):
        eng.features_tile(tile, ['no_such_layer'])
    eng.set_contents_and_styles([], [])
    with pytest.raises(StxError):
        eng.sc_grad_tile(tile, (0, 0), (0, 0), ['conv4_2'], [], {}, {'conv4_2': 1.0}, {})
