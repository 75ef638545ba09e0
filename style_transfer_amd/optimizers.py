"""Image optimizers on device-resident state (the reference's ``optimizers.py``).

``AdamOptimizer`` (optimizers.py:11-61) and ``LBFGSOptimizer`` (optimizers.py:64-138) keep the
same interface -- ``update(opfunc) -> (averaged_or_current_image, loss)``, ``roll(xy)``,
``set_params(last_iterate)`` -- but the arrays are ``DeviceArray`` on the master GPU and the
arithmetic is the fused stx_adam_step / stx_vec_* kernels.  The state is stored UN-rolled, so
``roll`` only tracks the accumulated shift: every operation here is elementwise or a dot
product and therefore commutes with the circular shift the reference applies to its copies.
"""

import numpy as np

from . import image_ops
from .resample import BILINEAR, LANCZOS, resample_device


class _Ewma:
    """State of average.EWMA (beta, beta_accum) around a device array."""

    def __init__(self, engine, shape, beta, correct_bias=True):
        self.beta = beta
        self.beta_accum = 1.0 if correct_bias else 0.0
        self.value = engine.empty(shape).zero()

    def advance(self):
        """Bias-correction denominator AFTER this update: 1 - beta_accum * beta."""
        self.beta_accum *= self.beta
        return 1 - self.beta_accum


class AdamOptimizer:
    """Adam with step-size decay and iterate averaging (optimizers.py:11-61)."""

    def __init__(self, engine, params, step_size=1, b1=0.9, b2=0.999, bp1=0, decay=0, power=1,
                 biased_g1=False):
        self.engine = engine
        self.params = params
        self.step_size, self.decay, self.power = step_size, decay, power
        self.i = 1
        self.xy = np.zeros(2, np.int32)
        self.g1 = _Ewma(engine, params.shape, b1, correct_bias=not biased_g1)
        self.g2 = _Ewma(engine, params.shape, b2)
        self.p1 = _Ewma(engine, params.shape, bp1)
        self.avg = engine.empty(params.shape)

    def update(self, opfunc):
        lr = self.step_size / self.i ** self.power
        self.i += self.decay
        loss, grad = opfunc(self.params)
        c1, c2, cp = self.g1.advance(), self.g2.advance(), self.p1.advance()
        image_ops.adam_step(self.engine, self.params, grad, self.g1.value, self.g2.value,
                            self.p1.value, self.avg, lr, self.g1.beta, self.g2.beta, self.p1.beta,
                            c1, c2, cp)
        return self.avg, loss

    def roll(self, xy):
        self.xy += np.asarray(xy, np.int32)

    def set_params(self, last_iterate):
        """New scale: ``last_iterate`` is the resized image (DeviceArray); g1/p1 are Lanczos-
        resized and g2 bilinear-resized and clamped at 0 (optimizers.py:53-61)."""
        self.i = 1
        old = [self.params, self.avg]
        self.params = last_iterate
        hw = self.params.shape[-2:]
        for ew, method, clamp in ((self.g1, LANCZOS, False), (self.g2, BILINEAR, True),
                                  (self.p1, LANCZOS, False)):
            resized = resample_device(self.engine, ew.value, hw, method, clamp_min_zero=clamp)
            ew.value.free()
            ew.value = resized
        self.avg = self.engine.empty(self.params.shape)
        for a in old:
            if a is not self.params:
                a.free()


class LBFGSOptimizer:
    """L-BFGS with fixed-size steps, no line search (optimizers.py:64-138).

    The two-loop recursion never leaves the GPU: every dot product lands in a device scalar and
    the axpy / scale that consumes it reads the coefficient from there (stx_vec_*_dev), so a
    step costs one host synchronisation -- the curvature test ``s.y > 1e-10`` that decides
    whether the pair is kept -- instead of one per dot product (~40 at a full memory).
    Image-sized work arrays come from a pool and are reused across steps (raw device
    allocations synchronise the whole GPU)."""

    def __init__(self, engine, params, initial_step=0.1, n_corr=10):
        self.engine = engine
        self.params = params
        self.initial_step, self.n_corr = initial_step, n_corr
        self.xy = np.zeros(2, np.int32)
        self.loss, self.grad = None, None
        self.sk, self.yk, self.syk = [], [], []
        self._pool = []
        # slots 0..n_corr-1: the s_i . q of the first loop; n_corr: y.y / sum|s|; n_corr+1: y_i . q
        self._scalars = image_ops.DeviceScalars(engine, n_corr + 2)

    # ---- image-sized scratch arrays, reused
    def _take(self, like):
        for i, a in enumerate(self._pool):
            if a.shape == like.shape:
                return self._pool.pop(i)
        return self.engine.empty(like.shape)

    def _give(self, *arrays):
        self._pool.extend(arrays)

    def _drop_pool(self):
        for a in self._pool:
            a.free()
        self._pool = []

    def _copy(self, src):
        return self._take(src).copy_from(src)

    def update(self, opfunc):
        eng = self.engine
        if self.loss is None:
            self.loss, grad = opfunc(self.params)
            self.grad = self._copy(grad)
        s = self.inv_hv(self.grad)
        image_ops.scale(eng, -1.0, s)
        if not self.sk:
            # s *= initial_step / mean|s|
            image_ops.abs_sum_async(eng, s, self._scalars.ptr(self.n_corr))
            image_ops.scale_dev(eng, self.initial_step, self._scalars.ptr(self.n_corr), s,
                                den_div=s.size)
        elif len(self.sk) < self.n_corr:
            image_ops.scale(eng, len(self.sk) / self.n_corr, s)
        image_ops.axpy(eng, 1.0, s, self.params)
        loss, grad = opfunc(self.params)
        y = self._copy(grad)
        image_ops.axpy(eng, -1.0, self.grad, y)
        self.store_curvature_pair(s, y)
        self.loss = loss
        self.grad.copy_from(grad)
        return self.params, loss

    def store_curvature_pair(self, s, y):
        sy = image_ops.dot(self.engine, s, y)          # the step's one host synchronisation
        if sy > 1e-10:
            self.sk.append(s), self.yk.append(y), self.syk.append(sy)
        else:
            self._give(s, y)
        if len(self.sk) > self.n_corr:
            self._give(self.sk[0], self.yk[0])
            self.sk, self.yk, self.syk = self.sk[1:], self.yk[1:], self.syk[1:]

    def inv_hv(self, p):
        eng, sc = self.engine, self._scalars
        p = self._copy(p)
        m = len(self.sk)
        for j in range(m - 1, -1, -1):                  # newest to oldest
            # alpha_j = s_j . p / sy_j ;  p -= alpha_j y_j
            image_ops.dot_async(eng, self.sk[j], p, sc.ptr(j))
            image_ops.axpy_dev(eng, -1.0, sc.ptr(j), self.syk[j], self.yk[j], p)
        if m:
            y = self.yk[-1]
            image_ops.dot_async(eng, y, y, sc.ptr(self.n_corr))
            image_ops.scale_dev(eng, self.syk[-1], sc.ptr(self.n_corr), p)      # p *= sy / y.y
        for j in range(m):                              # oldest to newest
            # beta = y_j . p / sy_j ;  p += (alpha_j - beta) s_j
            image_ops.dot_async(eng, self.yk[j], p, sc.ptr(self.n_corr + 1))
            image_ops.axpy_dev(eng, 1.0, sc.ptr(j), self.syk[j], self.sk[j], p,
                               c2=-1.0, b_ptr=sc.ptr(self.n_corr + 1), db=self.syk[j])
        return p

    def roll(self, xy):
        self.xy += np.asarray(xy, np.int32)

    def set_params(self, last_iterate):
        """New scale: the memory is cleared (optimizers.py:134-138) and every array of the old
        size is released."""
        old = self.params
        self.params = last_iterate
        self.loss = None
        for a in self.sk + self.yk + ([self.grad] if self.grad is not None else []):
            a.free()
        self.grad = None
        self.sk, self.yk, self.syk = [], [], []
        self._drop_pool()
        if old is not None and old is not last_iterate:
            old.free()
