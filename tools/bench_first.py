"""Times the first layer's kernel alone through the C ABI's kernel hook (conv_first_kernel<false>:
convolution + bias + ReLU + stores, no Gram partials):   python tools/bench_first.py [size]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd import lib
from style_transfer_amd.engine import TileEngine
from style_transfer_amd.netspec import builtin_net

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = TileEngine(builtin_net('vgg19'), 0)
rng = np.random.RandomState(0)
x = eng.to_device(rng.uniform(-110, 120, (3, size, size)).astype(np.float32))
w = eng.to_device((rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32))
b = eng.to_device(rng.standard_normal(64).astype(np.float32))
y = eng.empty((64, size, size))
call = lambda: lib.call('stx_op_conv_forward', eng.handle, x.ptr, 3, size, size, w.ptr, b.ptr, 64, 3, 1, y.ptr)
for _ in range(5):
    call()
eng.sync()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    call()
eng.sync()
us = (time.perf_counter() - t0) / n * 1e6
print('first layer %dx%d (fused kernel %s): %.1f us per launch = %.2f TB/s of output'
      % (size, size, os.environ.get('STX_CONV_FIRST_FUSED', '1'), us, 64 * size * size * 4 / us / 1e6))
