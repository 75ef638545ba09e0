"""Per-kernel-group timing of one tile evaluation (HIP events inside the engine)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd.engine import TileEngine
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.weights import synthetic_weights

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = sys.argv[3] if len(sys.argv) > 3 else 'vgg19'
net = builtin_net(model)
eng = TileEngine(net, 0, synthetic_weights(net, 0))
rng = np.random.RandomState(0)
cl, sl = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
cw, sw = {'conv4_2': 0.05}, {l: 0.2 for l in sl}
eng.set_contents_and_styles(
    [{l: np.abs(rng.standard_normal(eng.feature_shape(l, size, size))).astype(np.float32) for l in cl}],
    [{l: np.tril(rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32) for l in sl}])
tile = eng.to_device(rng.uniform(-120, 120, (3, size, size)).astype(np.float32))
grad = eng.empty((3, size, size))
for _ in range(2):
    eng.sc_grad_tile_async(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw, grad_out=grad)
eng.sync()
eng.profile(True)
eng.clock_marks(True)      # the shader clock inside every 2-D Winograd launch (4th column)
acc = {}
order = []
for _ in range(reps):
    eng.sc_grad_tile_async(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw, grad_out=grad)
    for label, ms, flops, mhz in eng.profile_read(clock=True):
        if label not in acc:
            acc[label] = [0.0, flops, 0.0, 0]
            order.append(label)
        acc[label][0] += ms
        if mhz > 0:
            acc[label][2] += mhz
            acc[label][3] += 1
    eng.clock_marks_read()
tot_ms = tot_fl = 0
print('%-18s %9s %9s %8s %8s' % ('group', 'ms', 'GFLOP', 'TFLOP/s', 'MHz'))
for label in order:
    ms, fl = acc[label][0] / reps, acc[label][1]
    tot_ms += ms
    tot_fl += fl
    print('%-18s %9.3f %9.1f %8.1f %8s' % (label, ms, fl / 1e9, fl / ms / 1e9 if fl else 0,
                                           '%.0f' % (acc[label][2] / acc[label][3]) if acc[label][3] else ''))
print('%-18s %9.3f %9.1f %8.1f' % ('TOTAL (events)', tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9))
