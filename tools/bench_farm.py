"""Step time of the full optimizer iteration through TileFarm (multi-tile, multi-engine)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd import image_ops
from style_transfer_amd.farm import TileFarm, tile_grid
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.optimizers import AdamOptimizer
from style_transfer_amd.weights import synthetic_weights

size = int(sys.argv[1]); tile = int(sys.argv[2]); devices = [int(d) for d in sys.argv[3].split(',')]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
net = builtin_net('vgg19')
farm = TileFarm(net, devices, synthetic_weights(net, 0), verbose=False,
                streams_per_device=int(os.environ.get('STREAMS', '4')))
eng = farm.master
rng = np.random.RandomState(0)
cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
contents = [{l: np.abs(rng.standard_normal(eng.feature_shape(l, size, size))).astype(np.float32) for l in cl}]
styles = [{l: np.tril(rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}]
farm.set_contents_and_styles(contents, styles)
img = eng.to_device(rng.uniform(-120, 120, (3, size, size)).astype(np.float32))
grad = eng.empty((3, size, size)); old = eng.empty((3, size, size)).copy_from(img)
opt = AdamOptimizer(eng, img, step_size=15, bp1=0.95, decay=0.05, power=0.5)
mean = (103.939, 116.779, 123.68)
def opfunc(p):
    xy = np.int32(rng.uniform(-0.5, 0.5, 2) * size) // 8 * 8
    loss = farm.eval_sc_grad(p, grad, xy, cl, sl, {}, cw, sw, tile)
    reg = image_ops.regularizers(eng, p, grad, mean, 5.0, 2.0, 2.0, 6.0)
    eng.sync()
    return loss + reg.value, grad
for _ in range(3):
    avg, loss = opt.update(opfunc); image_ops.step_stats(eng, avg, old)
n0 = farm.tile_evals; t0 = time.perf_counter()
for _ in range(steps):
    avg, loss = opt.update(opfunc); image_ops.step_stats(eng, avg, old)
dt = time.perf_counter() - t0
n = farm.tile_evals - n0
print('size %d tile %d devices %s: %.2f ms/step, %d tiles/step, %.2f ms/tile-iteration, %.1f tile-it/s, loss %.5g'
      % (size, tile, devices, dt / steps * 1e3, n // steps, dt / n * 1e3, n / dt, loss))
farm.close()
