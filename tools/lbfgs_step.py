"""The L-BFGS step's own time (no tile evaluations): LBFGSOptimizer.update around a three-pass
quadratic on the device, fused passes against one launch per BLAS-1 call, at a full memory.

    python tools/lbfgs_step.py [size ...]        (default 2048 4096)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd import image_ops  # noqa: E402
from style_transfer_amd.engine import TileEngine  # noqa: E402
from style_transfer_amd.netspec import builtin_net  # noqa: E402
from style_transfer_amd.optimizers import LBFGSOptimizer  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096]
    eng = TileEngine(builtin_net('vgg19'), 0, None)
    for size in sizes:
        rng = np.random.RandomState(0)
        shape = (3, size, size)
        x0 = rng.uniform(-100, 100, shape).astype(np.float32)
        tgt = eng.to_device(rng.uniform(-100, 100, shape).astype(np.float32))
        # a diagonal quadratic with uneven curvature: its gradient 2 w (x - t)
        w = eng.to_device(np.exp(rng.uniform(-1, 1, shape)).astype(np.float32))
        diff = eng.empty(shape)

        def f(x):
            diff.copy_from(x)
            image_ops.axpy(eng, -1.0, tgt, diff)
            image_ops.scale(eng, 2.0, diff)
            return 0.0, diff

        def timed(n, fn):
            eng.sync()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            eng.sync()
            return (time.perf_counter() - t) / n * 1e3
        params = eng.to_device(x0)
        base = timed(20, lambda: f(params))
        line = 'size %4d: the quadratic alone %.3f ms;' % (size, base)
        for fused in ('1', '0'):
            os.environ['STX_LBFGS_FUSED'] = fused
            params = eng.to_device(x0)
            opt = LBFGSOptimizer(eng, params)
            for _ in range(14):
                opt.update(f)
            assert len(opt.sk) == opt.n_corr, len(opt.sk)
            ms = timed(20, lambda: opt.update(f))
            line += '  %s step %.3f ms (own %.3f)' % ('fused' if fused == '1' else 'unfused', ms, ms - base)
            opt.set_params(None)
        print(line, flush=True)
        for a in (tgt, w, diff):
            a.free()


if __name__ == '__main__':
    main()
