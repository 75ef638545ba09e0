"""The flip-accounting used by the GPU tile-path tests, checked on the CPU: perturb the oracle's
own activations at the float32 noise level (what a second fp32 implementation does), run the
oracle's backward pass on both sets, and require that every gradient difference above the
continuous tolerance lies inside the region tests/gpu_helpers.decision_taint marks."""

import numpy as np

from tests.gpu_helpers import decision_taint
from tests.helpers import DEFAULT_STYLE_LAYERS, make_oracle, normalized_weights


def test_taint_covers_every_discontinuous_difference():
    om, _ = make_oracle('vgg19')
    rng = np.random.RandomState(11)
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    full = rng.uniform(-110, 120, (3, 96, 120)).astype(np.float32)
    style = rng.uniform(-110, 120, (3, 48, 56)).astype(np.float32)
    om.styles = [om.style_grams([style], sl, 512)]
    om.contents = [om.prepare_features(full, cl, 512)]
    tile = np.ascontiguousarray(full[:, 8:88, 16:112])
    _, ref_grad = om.sc_grad_tile(tile, (8, 16), cl, sl, {}, cw, sw)
    blobs = om.blob_names[:om.blob_names.index('conv5_1') + 1]
    ref_acts = {b: om.net.blobs[b].data[0].copy() for b in blobs}
    # a second "implementation": same activations up to 3e-6 relative noise, which flips a
    # handful of near-tie ReLU signs and pooling winners
    acts = {}
    for b in blobs:
        noise = 1 + 3e-6 * rng.standard_normal(ref_acts[b].shape)
        bump = 1e-6 * np.abs(ref_acts[b]).max() * rng.standard_normal(ref_acts[b].shape)
        a = ref_acts[b] * noise + np.where(ref_acts[b] > 0, bump, 0)
        # pre-activation noise around zero: switch on a few elements that were clipped
        wake = (ref_acts[b] == 0) & (rng.uniform(size=a.shape) < 2e-4)
        a = np.where(wake, np.abs(bump), np.maximum(a, 0))
        acts[b] = a.astype(np.float32)
    _, same_grad = om.sc_grad_tile(tile, (8, 16), cl, sl, {}, cw, sw, activations=acts)
    taint, n_relu, n_pool = decision_taint(om.net.layers, acts, ref_acts, 'conv5_1',
                                           {'data': tile.shape})
    assert n_relu > 0 and n_pool > 0          # the perturbation did flip decisions
    assert 0 < taint.mean() <= 1
    diff = np.abs(np.float64(same_grad) - ref_grad).max(axis=0) / np.abs(ref_grad).max()
    clean = ~taint
    if clean.any():
        assert diff[clean].max() < 2e-5, diff[clean].max()
    # and the analysis is not vacuous: with identical activations nothing is tainted
    t0, r0, p0 = decision_taint(om.net.layers, ref_acts, ref_acts, 'conv5_1', {'data': tile.shape})
    assert not t0.any() and r0 == 0 and p0 == 0
