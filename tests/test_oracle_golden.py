"""The numpy oracle must reproduce what the reference's own Python produced
(tests/golden/reference_vectors.npz, made by tests/golden/make_golden.py)."""

import numpy as np
import pytest

from oracle import num_ops, optim
from oracle.tile_path import regularizer_loss_grad
from tests.helpers import (DEFAULT_STYLE_LAYERS, make_oracle, normalized_weights, rel_err,
                           u8_to_params)


def test_numeric_helpers(golden):
    g = golden
    feat = g['num.feat']
    assert rel_err(num_ops.gram_lower(feat), g['num.gram']) < 1e-5
    assert np.all(np.triu(g['num.gram'], 1) == 0)         # reference Gram is lower-triangular
    assert rel_err(num_ops.symm_lower_times(g['num.symm_in'], feat.reshape(64, -1)),
                   g['num.symm_out']) < 1e-5
    assert num_ops.half_sq_norm(g['num.symm_in']) == pytest.approx(float(g['num.norm2']), rel=1e-5)
    assert rel_err(num_ops.l1_normalize(feat.copy()), g['num.normalize']) < 1e-5
    img = g['num.img']
    for beta in (2, 1.5):
        loss, grad = num_ops.tv_loss_grad(img / np.float32(127.5), beta)
        assert loss == pytest.approx(float(g['num.tv_loss_%g' % beta]), rel=1e-5)
        assert rel_err(grad, g['num.tv_grad_%g' % beta]) < 1e-5
    loss, grad = num_ops.p_norm_loss_grad(img / np.float32(127.5), 6.0)
    assert loss == pytest.approx(float(g['num.p6_loss']), rel=1e-5)
    assert rel_err(grad, g['num.p6_grad']) < 1e-5
    assert np.array_equal(num_ops.roll_xy(img.copy(), (3, -5)), g['num.roll_3_-5'])


@pytest.mark.parametrize('tag,model', [('vgg19', 'vgg19'), ('vgg16avg', 'vgg16_avgpool')])
def test_tile_path(golden, tag, model):
    g = {k[len('tile.%s.' % tag):]: v for k, v in golden.items() if k.startswith('tile.%s.' % tag)}
    om, _ = make_oracle(model)
    tile_size = int(g['tile_size'])
    content_layers, content_weight = normalized_weights(['conv4_2'], 0.05)
    style_layers, style_weight = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    styles = [u8_to_params(g[k]) for k in sorted(g) if k.startswith('style') and k.endswith('_u8')]
    np.random.seed(123)
    om.styles = [om.style_grams(styles, style_layers, tile_size)]
    om.contents = [om.prepare_features(u8_to_params(g['content_u8']), content_layers, tile_size)]
    for layer in style_layers:
        gram = om.styles[0][layer]
        assert gram.sum(dtype=np.float64) == pytest.approx(float(g['gram_sum.' + layer]), rel=1e-4)
        assert rel_err(np.diag(gram)[:8], g['gram_diag8.' + layer]) < 1e-4
    cf = om.contents[0]['conv4_2']
    assert tuple(cf.shape) == tuple(g['content_feat_shape'])
    assert rel_err(cf[0], g['content_feat_c0']) < 1e-4
    lw = {'conv3_1': float(g['lw_conv3_1'])}
    loss, grad = om.sc_grad(g['img_rolled'], g['roll'], tile_size, content_layers, style_layers,
                            lw, content_weight, style_weight)
    assert loss == pytest.approx(float(g['loss']), rel=1e-4)
    assert rel_err(grad, g['grad']) < 1e-4
    tile = g['img_rolled'][:, 8:48, 16:72]
    tloss, tgrad = om.sc_grad_tile(tile, (8, 16), content_layers, style_layers, lw,
                                   content_weight, style_weight)
    assert tloss == pytest.approx(float(g['single.loss']), rel=1e-4)
    assert rel_err(tgrad, g['single.grad']) < 1e-4
    feats = om.features_tile(tile, ['pool1', 'conv5_1'])
    assert rel_err(feats['conv5_1'], g['single.feat_conv5_1']) < 1e-4
    assert feats['pool1'].sum(dtype=np.float64) == pytest.approx(float(g['single.feat_pool1_sum']),
                                                                 rel=1e-4)


def _quad(target):
    def f(x):
        d = x - target
        return float(np.sum(d * d, dtype=np.float64)), (2 * d).astype(np.float32)
    return f


@pytest.mark.parametrize('biased', [0, 1])
def test_adam_trajectory(golden, biased):
    params, tgt = golden['opt.x0'].copy(), golden['opt.target'].copy()
    opt = optim.Adam(params, step_size=15, bp1=1 - 1 / 20, decay=0.05, power=0.5,
                     biased_g1=bool(biased))
    for i, xy in enumerate(golden['opt.rolls']):
        num_ops.roll_xy(params, xy), num_ops.roll_xy(tgt, xy)
        opt.roll(xy)
        avg, loss = opt.update(_quad(tgt))
        num_ops.roll_xy(params, -xy), num_ops.roll_xy(tgt, -xy)
        opt.roll(-xy)
        assert rel_err(avg, golden['opt.adam_biased%d.avg' % biased][i]) < 1e-5
        assert loss == pytest.approx(golden['opt.adam_biased%d.loss' % biased][i], rel=1e-5)
    assert rel_err(params, golden['opt.adam_biased%d.params' % biased]) < 1e-5


def test_lbfgs_trajectory(golden):
    params, tgt, scale = golden['opt.x0'].copy(), golden['opt.target'], golden['opt.lbfgs.scale']

    def f(x):
        d = (x - tgt) * scale
        return float(np.sum(d * d, dtype=np.float64)), (2 * d * scale).astype(np.float32)
    opt = optim.Lbfgs(params)
    for i in range(len(golden['opt.lbfgs.loss'])):
        p, loss = opt.update(f)
        assert rel_err(p, golden['opt.lbfgs.params'][i]) < 2e-4
        assert loss == pytest.approx(golden['opt.lbfgs.loss'][i], rel=2e-3, abs=1e-3)


def test_regularizers_are_additive(golden):
    img = golden['num.img']
    grad = np.zeros_like(img)
    mean = np.float32((103.939, 116.779, 123.68)).reshape(3, 1, 1)
    loss = regularizer_loss_grad(img, mean, grad)
    tv_l, tv_g = num_ops.tv_loss_grad(img / np.float32(127.5), 2.0)
    p_l, p_g = num_ops.p_norm_loss_grad((img + mean - np.float32(127.5)) / np.float32(127.5), 6.0)
    assert loss == pytest.approx(5 * tv_l + 2 * p_l, rel=1e-6)
    assert rel_err(grad, 5 * tv_g + 2 * p_g) < 1e-6


def test_lbfgs_fixture_outcomes_of_the_reference(golden):
    """tests/golden/lbfgs_branches.npz (tests/golden/branch_sets.py lbfgs: the reference's own code on the
    e2e_lbfgs fixture with its convolutions / Gram matrices computed by other float32 implementations):
    the first outcome IS the committed fixture (and the one nearly all runs reproduce), the others are
    distinct pictures that start from the same objective."""
    from tests.helpers import lbfgs_reference_outcomes
    outs = lbfgs_reference_outcomes(golden)
    assert len(outs) >= 3 and sum(o['runs'] for o in outs) >= 30
    assert np.array_equal(outs[0]['log'], golden['e2e_lbfgs.log']) and outs[0]['runs'] >= 25
    for i, o in enumerate(outs[1:], 1):
        assert o['log'].shape == outs[0]['log'].shape and o['final_raw'].shape == outs[0]['final_raw'].shape
        rel = np.abs(o['log'][:, 2] / outs[0]['log'][:, 2] - 1)
        assert rel[0] < 1e-6 and rel.max() < 1e-2
        for q in outs[:i]:                       # pairwise distinct pictures
            assert np.abs(o['final_raw'] - q['final_raw']).max() >= 0.05


def test_config4_miniature_branches_of_the_reference(golden):
    """tests/golden/cfg4_branches.npz (tests/golden/branch_sets.py cfg4: the reference's own code on the
    config-4 miniature with its convolutions computed by other float32 implementations): the first
    branch IS the committed fixture's trajectory, the others leave it by more than the 2e-4 band from the
    second step on, and start from the same objective; the picture of a raw array is the reference's."""
    from tests.helpers import cfg4_reference_branches, matching_branch, raw_to_u8
    assert np.array_equal(raw_to_u8(golden['e2e_cfg4.final_raw']), golden['e2e_cfg4.final_u8'])
    branches = cfg4_reference_branches(golden)
    assert len(branches) == 2 and sum(b['runs'] for b in branches) >= 20
    assert branches[0]['runs'] >= 10 and branches[1]['runs'] >= 3    # both are trajectories several implementations take
    assert matching_branch(branches, golden['e2e_cfg4.log'][:, 2]) is branches[0]
    for b in branches[1:]:
        assert b['log'].shape == branches[0]['log'].shape and b['final_raw'].shape == branches[0]['final_raw'].shape
        rel = np.abs(b['log'][:, 2] / branches[0]['log'][:, 2] - 1)
        assert rel[0] < 1e-6 and rel[1:].max() > 2e-4 and rel.max() < 1e-2


def test_stable_lbfgs_fixture_has_one_trajectory(golden):
    """tests/golden/stable_runs.json (tests/golden/branch_sets.py stable): the reference's own code on the
    e2e_stable fixture (make_golden.py 4j) with its Convolution layer computed by 14 other float32
    implementations and under calibrated noise -- every run within a quarter of the 2e-4 band the GPU test
    holds the shipped kernels to, at every step."""
    import json
    import os
    runs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stable_runs.json')))['runs']
    assert len(runs) >= 20 and {'torch', 'taps_rev', 'chunk16', 'pairwise8'} <= {r['run'] for r in runs}
    assert max(max(r['loss_rel']) for r in runs) < 6e-5
    assert golden['e2e_stable.log'].shape == (5, 4) and golden['e2e_stable.final_raw'].shape == (3, 275, 280)
