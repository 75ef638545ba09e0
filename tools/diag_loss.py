"""Where does the loss of a large tile differ from the oracle's?  Prints every term in the
oracle's float32 arithmetic and in float64 (from the same activations) beside the GPU's."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_helpers import gpu_engine, loss_from_activations
from tests.helpers import DEFAULT_STYLE_LAYERS, make_oracle, normalized_weights
from tests.test_gpu_fullsize import _random_targets, _smooth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
om, _ = make_oracle('vgg19')
eng = gpu_engine('vgg19')
rng = np.random.RandomState(2 * size)
cl, cw = normalized_weights(['conv4_2'], 0.05)
sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
om.contents, om.styles = _random_targets(om, rng, (2 * size, 2 * size), cl, sl)
eng.set_contents_and_styles(om.contents, om.styles)
tile = _smooth(rng, size, size)
start, roll = (size, 0), (-312, 200)
loss, grad = eng.sc_grad_tile(tile, start, roll, cl, sl, {}, cw, sw)
blobs = om.blob_names[:om.blob_names.index('conv5_1') + 1]
acts = eng.features_tile(tile, blobs)
om.roll_contents(roll)
ref_loss, _ = om.sc_grad_tile(tile, start, cl, sl, {}, cw, sw)
ref_acts = {b: om.net.blobs[b].data[0].copy() for b in blobs}
for name, a in (('oracle acts', ref_acts), ('gpu acts', acts)):
    for dt in (np.float32, np.float64):
        total, terms = loss_from_activations(om, a, start, cl, sl, {}, cw, sw, dt)
        print('%-12s %-8s total %.9e  %s' % (name, dt.__name__, total,
              ' '.join('%s=%.6e' % kv for kv in terms.items())))
print('gpu loss     %.9e\noracle loss  %.9e   rel diff %.3e' % (loss, ref_loss, abs(loss - ref_loss) / ref_loss))
