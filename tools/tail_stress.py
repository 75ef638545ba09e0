"""Run-to-run determinism of the tile path on small odd tiles: the number of distinct gradients in
N evaluations of the same tile must be 1.  (Written for an experiment that cut the work items of
a launch's last, mostly empty round into K slices whose partial sums met in a scratch buffer
inside the running kernel -- DESIGN.md section 7: that hand-over was not reliable and showed up
here as 2 .. 150 distinct results; split-K slices, Gram partials and every reduction of the
shipped kernels are added in a fixed order.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd.engine import TileEngine
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.weights import synthetic_weights

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
net = builtin_net('vgg19')
eng = TileEngine(net, 0, synthetic_weights(net, 0))
rng = np.random.RandomState(0)
cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
bad_total = 0
for th, tw in ((64, 80), (37, 53), (96, 96), (33, 130), (50, 44), (181, 181), (362, 362)):
    eng.set_contents_and_styles(
        [{l: np.abs(rng.standard_normal(eng.feature_shape(l, th, tw))).astype(np.float32) for l in cl}],
        [{l: np.tril(rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}])
    tile = eng.to_device(rng.uniform(-120, 120, (3, th, tw)).astype(np.float32))
    grad = eng.empty((3, th, tw))
    import collections, hashlib
    seen = collections.Counter()
    order = []
    for r in range(reps):
        p = eng.sc_grad_tile_async(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw, grad_out=grad)
        eng.sync()
        h = hashlib.md5(grad.get().tobytes()).hexdigest()[:6]
        if h not in seen:
            order.append(h)
        seen[h] += 1
    bad = reps - max(seen.values())
    print('%dx%d: %d distinct results in %d repetitions; counts in order of appearance: %s'
          % (th, tw, len(seen), reps, [seen[h] for h in order][:12]))
    bad_total += bad
sys.exit(1 if bad_total else 0)
