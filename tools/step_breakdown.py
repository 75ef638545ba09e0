import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import image_ops
from style_transfer_amd.engine import TileEngine
from style_transfer_amd.farm import TileFarm, tile_grid
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.weights import synthetic_weights
net = builtin_net('vgg19'); w = synthetic_weights(net, 0)
eng = TileEngine(net, 0, w)
H = W = 2048
rng = np.random.RandomState(0)
img = eng.to_device(rng.uniform(-100, 100, (3, H, W)).astype(np.float32))
grad = eng.empty((3, H, W)); old = eng.empty((3, H, W)).copy_from(img)
tiles = [eng.empty((3, 1024, 1024)) for _ in range(4)]
rects = tile_grid((H, W), 1024)
MEAN = (103.939, 116.779, 123.68)
def t(f, n=50):
    f(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    eng.sync()
    return (time.perf_counter() - t0) / n * 1e3
print('cut x4 + sync      %.3f ms' % t(lambda: ([image_ops.cut_tile(eng, img, (8, 16), r, tl) for r, tl in zip(rects, tiles)], eng.sync())))
print('put x4             %.3f ms' % t(lambda: [image_ops.put_tile(eng, grad, (8, 16), r, tl) for r, tl in zip(rects, tiles)]))
print('regularizers+sync  %.3f ms' % t(lambda: (image_ops.regularizers(eng, img, grad, MEAN, 5.0, 2.0, 2.0, 6.0), eng.sync())))
from style_transfer_amd.optimizers import AdamOptimizer
opt = AdamOptimizer(eng, img, step_size=15, bp1=0.95, decay=0.05, power=0.5)
print('adam update        %.3f ms' % t(lambda: opt.update(lambda p: (1.0, grad))))
print('step_stats         %.3f ms' % t(lambda: image_ops.step_stats(eng, img, old)))
