"""Times the backward pass into the image (conv3x3_m4_kernel, 64 -> 3 channels) alone, through the
C ABI's kernel hook:  python tools/bench_small.py [size]     (STX_SMALL_TUNE=<KC*100+PR> with a
library built with STX_HIPCC_EXTRA=-DSTX_SMALL_SWEEP selects another tiling; same results)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd import lib
from style_transfer_amd.engine import TileEngine
from style_transfer_amd.netspec import builtin_net

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = TileEngine(builtin_net('vgg19'), 0)
rng = np.random.RandomState(0)
dy = eng.to_device(rng.standard_normal((64, size, size)).astype(np.float32))
w = eng.to_device((rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32))
dx = eng.empty((3, size, size))
call = lambda: lib.call('stx_op_conv_backward_data', eng.handle, dy.ptr, 64, size, size, w.ptr, 3, 3, None, dx.ptr)
for _ in range(5):
    call()
eng.sync()
ref = dx.get().copy()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    call()
eng.sync()
us = (time.perf_counter() - t0) / n * 1e6
print('%s: %.1f us per launch incl. the weight pack (%.2f TB/s of input), checksum %.6e'
      % (os.environ.get('STX_SMALL_TUNE', 'default'), us, 64 * size * size * 4 / us / 1e6, float(np.abs(ref).sum())))
