"""Numpy restatement of the reference's image optimizers (``optimizers.py``).

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``Ewma`` restates the un-vendored
``average.EWMA`` (requirements.txt:3; call sites optimizers.py:22-24,35-37,41-42):
``v <- beta*v + (1-beta)*x`` with ``get() = v / (1 - beta^t)`` when bias-corrected
("parity unpinned": the package is absent from /root/reference and this container).
Resizing between scales (optimizers.py:53-61) needs Pillow and is exercised through the
host-side tests, not here.
"""

import numpy as np

from .num_ops import EPS, roll_xy


class Ewma:
    def __init__(self, like, beta, correct_bias=True):
        self.beta = beta
        self.beta_accum = 1.0 if correct_bias else 0.0
        self.value = np.zeros_like(like)

    def update(self, x):
        self.beta_accum *= self.beta
        self.value *= self.beta
        self.value += (1 - self.beta) * x

    def get(self):
        return self.value / (1 - self.beta_accum)


class Adam:
    """optimizers.py:11-51: Adam with step decay and Polyak-style iterate averaging."""

    def __init__(self, params, step_size=1, b1=0.9, b2=0.999, bp1=0, decay=0, power=1,
                 biased_g1=False):
        self.params, self.step_size, self.decay, self.power = params, step_size, decay, power
        self.i = 1
        self.xy = np.zeros(2, np.int32)
        self.g1 = Ewma(params, b1, correct_bias=not biased_g1)
        self.g2 = Ewma(params, b2)
        self.p1 = Ewma(params, bp1)

    def update(self, opfunc):
        lr = self.step_size / self.i ** self.power
        self.i += self.decay
        loss, grad = opfunc(self.params)
        self.g1.update(grad)
        self.g2.update(grad ** 2)
        step = self.g1.get() / (np.sqrt(self.g2.get()) + EPS)
        self.params += np.float32(-lr) * step.astype(np.float32)
        self.p1.update(self.params)
        return roll_xy(self.p1.get(), -self.xy), loss

    def roll(self, xy):
        xy = np.asarray(xy)
        if (xy == 0).all():
            return
        self.xy += xy
        for ew in (self.g1, self.g2, self.p1):
            roll_xy(ew.value, xy)


class Lbfgs:
    """optimizers.py:64-138: L-BFGS two-loop recursion with fixed-size steps."""

    def __init__(self, params, initial_step=0.1, n_corr=10):
        self.params, self.initial_step, self.n_corr = params, initial_step, n_corr
        self.xy = np.zeros(2, np.int32)
        self.loss = self.grad = None
        self.sk, self.yk, self.syk = [], [], []

    @staticmethod
    def _dot(a, b):
        return float(np.dot(a.ravel(), b.ravel()))

    def inv_hv(self, p):
        p = p.copy()
        alphas = []
        for s, y, sy in zip(self.sk[::-1], self.yk[::-1], self.syk[::-1]):
            alphas.append(self._dot(s, p) / sy)
            p += np.float32(-alphas[-1]) * y
        if self.sk:
            p *= self.syk[-1] / self._dot(self.yk[-1], self.yk[-1])
        for s, y, sy, alpha in zip(self.sk, self.yk, self.syk, alphas[::-1]):
            beta = self._dot(y, p) / sy
            p += np.float32(alpha - beta) * s
        return p

    def update(self, opfunc):
        if self.loss is None:
            self.loss, self.grad = opfunc(self.params)
        s = -self.inv_hv(self.grad)
        if not self.sk:
            s *= self.initial_step / np.mean(abs(s))
        elif len(self.sk) < self.n_corr:
            s *= len(self.sk) / self.n_corr
        self.params += s
        loss, grad = opfunc(self.params)
        y = grad - self.grad
        sy = self._dot(s, y)
        if sy > 1e-10:
            self.sk.append(s), self.yk.append(y), self.syk.append(sy)
        if len(self.sk) > self.n_corr:
            self.sk, self.yk, self.syk = self.sk[1:], self.yk[1:], self.syk[1:]
        self.loss, self.grad = loss, grad
        return self.params, loss

    def roll(self, xy):
        xy = np.asarray(xy)
        if (xy == 0).all():
            return
        self.xy += xy
        if self.grad is not None:
            roll_xy(self.grad, xy)
        for s, y in zip(self.sk, self.yk):
            roll_xy(s, xy)
            roll_xy(y, xy)
