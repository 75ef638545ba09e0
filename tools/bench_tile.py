"""Times stx_sc_grad_tile on a device-resident tile (quick kernel-level benchmark)."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd.engine import TileEngine
from style_transfer_amd.netspec import builtin_net
from oracle.caffe_net import synthetic_weights

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = sys.argv[3] if len(sys.argv) > 3 else 'vgg19'
net = builtin_net(model)
eng = TileEngine(net, 0, synthetic_weights(net.as_dicts(), 0))
rng = np.random.RandomState(0)
cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
contents = [{l: np.abs(rng.standard_normal(eng.feature_shape(l, size, size))).astype(np.float32) for l in cl}]
styles = [{l: np.tril(rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}]
eng.set_contents_and_styles(contents, styles)
tile = eng.to_device(rng.uniform(-120, 120, (3, size, size)).astype(np.float32))
grad = eng.empty((3, size, size))
for _ in range(2):
    p = eng.sc_grad_tile_async(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw, grad_out=grad)
    eng.sync()
t0 = time.perf_counter()
for _ in range(iters):
    p = eng.sc_grad_tile_async(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw, grad_out=grad)
eng.sync()
dt = (time.perf_counter() - t0) / iters
flop = {'vgg19': 1514240, 'vgg16': 1219328}[model.split('_')[0]] * size * size
print('size %d: %.3f ms / tile-iteration (event %.3f ms), %.1f TFLOP/s algorithmic, loss %.6g'
      % (size, dt * 1e3, eng.last_tile_ms(), flop / dt / 1e12, p.loss))
