"""Per-scale cost of the step loop: ms per optimizer step at the single-tile and multi-tile
scales of the `--size 2048 --tile-size 1024` pyramid, through TileFarm (the product path), with
the GPU span of the tile evaluation (HIP events) and the host's queueing cost beside it.

    python tools/scale_steps.py [sizes...]        STX_RUN_AHEAD=0: the blocking loop (every step collected
                                                  before the next is queued) instead of the product's run-ahead loop
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd import image_ops, lib
from style_transfer_amd.farm import TileFarm
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.optimizers import AdamOptimizer
from style_transfer_amd.weights import synthetic_weights

MEAN = (103.939, 116.779, 123.68)
CL, SL = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
CW, SW = {'conv4_2': 0.05}, {l: 0.2 for l in SL}
sizes = [int(a) for a in sys.argv[1:]] or [256, 362, 512, 724, 1024, 1448, 2048]
net = builtin_net('vgg19')
farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
eng = farm.master
rng = np.random.RandomState(0)
for size in sizes:
    img_host = rng.uniform(-110, 120, (3, size, size)).astype(np.float32)
    contents = [farm.prepare_features_device(img_host, CL, 1024, passes=1)]
    feats = farm.prepare_features_device(img_host[:, :min(size, 512), :min(size, 512)], SL, 1024, passes=1)
    farm.set_contents_and_styles(contents, [{l: farm.gram_matrix(f) for l, f in feats.items()}])
    img = eng.to_device(img_host)
    grad, old = eng.empty(img.shape), eng.empty(img.shape).copy_from(img)
    opt = AdamOptimizer(eng, img, step_size=15, bp1=0.95, decay=0.05, power=0.5)
    st = np.random.RandomState(1)

    ahead = os.environ.get('STX_RUN_AHEAD', '1') != '0'
    in_flight = []

    def step():
        roll = np.int32(st.uniform(-0.5, 0.5, size=2) * size) // 8 * 8

        def opfunc(p):
            loss = farm.eval_sc_grad(p, grad, roll, CL, SL, {}, CW, SW, 1024, lazy=True)
            loss.add(image_ops.regularizers(eng, p, grad, MEAN, 5.0, 2.0, 2.0, 6.0), eng)
            return loss, grad
        avg, loss = opt.update(opfunc)
        if not ahead:
            image_ops.step_stats(eng, avg, old)
            return float(loss)
        # StyleTransfer.transfer's loop: queue this step, then collect the previous one
        stats = image_ops.step_stats_async(eng, avg, old)
        loss.seal(also=[eng])
        if in_flight:
            l, s_ = in_flight.pop()
            float(l), s_.values()
        in_flight.append((loss, stats))

    def drain():
        while in_flight:
            l, s_ = in_flight.pop()
            float(l), s_.values()

    for _ in range(5):
        step()
    drain()
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    drain()
    dt = (time.perf_counter() - t0) / n * 1e3
    tiles = farm.tile_evals
    gpu = max(e.last_tile_ms() for e in farm.engines[:max(1, min(4, ((size - 1) // 1024 + 1) ** 2))])
    # host cost of queueing one step without waiting for it
    t1 = time.perf_counter()
    roll = np.int32([8, 16])
    for _ in range(20):
        farm.eval_sc_grad(img, grad, roll, CL, SL, {}, CW, SW, 1024, lazy=True)
    queue = (time.perf_counter() - t1) / 20 * 1e3
    eng.sync()
    print('size %4d: %7.3f ms/step, tile span %7.3f ms, queueing eval_sc_grad %6.3f ms; tile evaluations %d'
          % (size, dt, gpu, queue, sum(e.query(lib.Q_TILE_EVALS) for e in farm.engines)), flush=True)
    for a in (img, grad, old):
        a.free()
farm.close()
