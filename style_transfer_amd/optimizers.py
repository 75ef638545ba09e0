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
    """L-BFGS with fixed-size steps, no line search (optimizers.py:64-138)."""

    def __init__(self, engine, params, initial_step=0.1, n_corr=10):
        self.engine = engine
        self.params = params
        self.initial_step, self.n_corr = initial_step, n_corr
        self.xy = np.zeros(2, np.int32)
        self.loss, self.grad = None, None
        self.sk, self.yk, self.syk = [], [], []

    def _copy(self, src):
        return self.engine.empty(src.shape).copy_from(src)

    def update(self, opfunc):
        eng = self.engine
        if self.loss is None:
            self.loss, grad = opfunc(self.params)
            self.grad = self._copy(grad)
        s = self.inv_hv(self.grad)
        image_ops.scale(eng, -1.0, s)
        if not self.sk:
            image_ops.scale(eng, self.initial_step / image_ops.mean_abs(eng, s), s)
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
        sy = image_ops.dot(self.engine, s, y)
        if sy > 1e-10:
            self.sk.append(s), self.yk.append(y), self.syk.append(sy)
        else:
            s.free(), y.free()
        if len(self.sk) > self.n_corr:
            self.sk[0].free(), self.yk[0].free()
            self.sk, self.yk, self.syk = self.sk[1:], self.yk[1:], self.syk[1:]

    def inv_hv(self, p):
        eng = self.engine
        p = self._copy(p)
        alphas = []
        for s, y, sy in zip(self.sk[::-1], self.yk[::-1], self.syk[::-1]):
            alphas.append(image_ops.dot(eng, s, p) / sy)
            image_ops.axpy(eng, -alphas[-1], y, p)
        if self.sk:
            y = self.yk[-1]
            image_ops.scale(eng, self.syk[-1] / image_ops.dot(eng, y, y), p)
        for s, y, sy, alpha in zip(self.sk, self.yk, self.syk, alphas[::-1]):
            beta = image_ops.dot(eng, y, p) / sy
            image_ops.axpy(eng, alpha - beta, s, p)
        return p

    def roll(self, xy):
        self.xy += np.asarray(xy, np.int32)

    def set_params(self, last_iterate):
        self.params = last_iterate
        self.loss, self.grad = None, None
        for a in self.sk + self.yk:
            a.free()
        self.sk, self.yk, self.syk = [], [], []
