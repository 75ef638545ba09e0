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
    allocations synchronise the whole GPU).

    Round 5: the passes are fused (``STX_LBFGS_FUSED=0`` keeps one launch per BLAS-1 call) --
    the axpy of one iteration of either loop runs with the dot product of the next, the first
    loop's last axpy with the scaling and the second loop's first dot product, ``y = g - g_old``
    with both of its dot products and the copy of the gradient, the step's two scalings with its
    addition to the image.  Every value is computed by the same float operations in the same
    order (the dot products by the same threads in the same grid), so the trajectory is the
    unfused one bit for bit; the step moves 89 array passes instead of 119 at a full memory."""

    def __init__(self, engine, params, initial_step=0.1, n_corr=10):
        import os
        self.engine = engine
        self.params = params
        self.initial_step, self.n_corr = initial_step, n_corr
        self.xy = np.zeros(2, np.int32)
        self.loss, self.grad = None, None
        self.sk, self.yk, self.syk = [], [], []
        self._pool = []
        self.fused = os.environ.get('STX_LBFGS_FUSED', '1') != '0'
        # slots 0..n_corr-1: the s_i . q of the first loop; n_corr: y.y / sum|s|; n_corr+1, +2: y_i . q
        # (alternating); n_corr+3 .. +6: two {s.y, y.y} pairs -- the newest kept pair's and the candidate's
        self._scalars = image_ops.DeviceScalars(engine, n_corr + 7)
        self._yy_slot = n_corr + 4          # where <y, y> of the newest kept pair lives
        self._pair_out = n_corr + 3         # where the next candidate pair's two products go

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
        if not self.sk:
            image_ops.scale(eng, -1.0, s)
            # s *= initial_step / mean|s|
            image_ops.abs_sum_async(eng, s, self._scalars.ptr(self.n_corr))
            image_ops.scale_dev(eng, self.initial_step, self._scalars.ptr(self.n_corr), s,
                                den_div=s.size)
            image_ops.axpy(eng, 1.0, s, self.params)
        elif self.fused:
            # s = (len / n_corr) * (-1 * s); params += s   (a full memory: times 1.0, exact)
            c2 = len(self.sk) / self.n_corr if len(self.sk) < self.n_corr else 1.0
            image_ops.scale2_axpy(eng, -1.0, c2, s, self.params)
        else:
            image_ops.scale(eng, -1.0, s)
            if len(self.sk) < self.n_corr:
                image_ops.scale(eng, len(self.sk) / self.n_corr, s)
            image_ops.axpy(eng, 1.0, s, self.params)
        loss, grad = opfunc(self.params)
        if self.fused:
            # y = grad - self.grad, self.grad = grad, <s, y> (the step's one host synchronisation), <y, y>
            y = self._take(grad)
            sy = image_ops.lbfgs_pair(eng, grad, self.grad, s, y, self._scalars.ptr(self._pair_out))
            self.store_curvature_pair(s, y, sy)
        else:
            y = self._copy(grad)
            image_ops.axpy(eng, -1.0, self.grad, y)
            self.store_curvature_pair(s, y)
            self.grad.copy_from(grad)
        self.loss = loss
        return self.params, loss

    def store_curvature_pair(self, s, y, sy=None):
        if sy is None:
            sy = image_ops.dot(self.engine, s, y)      # the step's one host synchronisation
        if sy > 1e-10:
            self.sk.append(s), self.yk.append(y), self.syk.append(sy)
            if self.fused:      # the candidate's <y, y> is the newest pair's now; the other two slots are free
                self._yy_slot = self._pair_out + 1
                self._pair_out = self.n_corr + 3 + (self._pair_out - self.n_corr - 3 + 2) % 4
        else:
            self._give(s, y)
        if len(self.sk) > self.n_corr:
            self._give(self.sk[0], self.yk[0])
            self.sk, self.yk, self.syk = self.sk[1:], self.yk[1:], self.syk[1:]

    def inv_hv(self, p):
        if self.fused:
            return self._inv_hv_fused(p)
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

    def _inv_hv_fused(self, g):
        """The same recursion, one pass per iteration: m + 1 launches for the first loop (its first dot
        product reads ``g`` itself, its first axpy writes the work array: no copy), m for the second."""
        eng, sc, m = self.engine, self._scalars, len(self.sk)
        p = self._take(g)
        if not m:
            return p.copy_from(g)
        sk, yk, syk = self.sk, self.yk, self.syk
        beta = (self.n_corr + 1, self.n_corr + 2)
        image_ops.dot_async(eng, sk[m - 1], g, sc.ptr(m - 1))
        src = g
        for j in range(m - 1, 0, -1):
            # p = src - alpha_j y_j ;  <s_{j-1}, p>
            image_ops.axpy_dot_dev(eng, -1.0, sc.ptr(j), syk[j], yk[j], p, sk[j - 1], sc.ptr(j - 1), src=src)
            src = p
        # p = (sy / y.y) (src - alpha_0 y_0) ;  <y_0, p>
        image_ops.axpy_dot_dev(eng, -1.0, sc.ptr(0), syk[0], yk[0], p, yk[0], sc.ptr(beta[0]), src=src,
                               scale_c=syk[-1], scale_den_ptr=sc.ptr(self._yy_slot))
        for j in range(m - 1):
            # p += (alpha_j - beta_j) s_j ;  <y_{j+1}, p>
            image_ops.axpy_dot_dev(eng, 1.0, sc.ptr(j), syk[j], sk[j], p, yk[j + 1], sc.ptr(beta[(j + 1) % 2]),
                                   c2=-1.0, b_ptr=sc.ptr(beta[j % 2]), db=syk[j])
        j = m - 1
        image_ops.axpy_dev(eng, 1.0, sc.ptr(j), syk[j], sk[j], p, c2=-1.0, b_ptr=sc.ptr(beta[j % 2]), db=syk[j])
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
