"""Full-image kernels (regularizers, Adam, L-BFGS algebra, cut/put, statistics, uint8) against
vectors produced by the reference's num_utils / optimizers and against the oracle."""

import numpy as np
import pytest

from oracle import num_ops
from oracle.tile_path import regularizer_loss_grad
from style_transfer_amd import image_ops
from style_transfer_amd.optimizers import AdamOptimizer, LBFGSOptimizer
from tests.gpu_helpers import gpu_engine, max_rel

pytestmark = pytest.mark.gpu
MEAN = np.float32((103.939, 116.779, 123.68)).reshape(3, 1, 1)


@pytest.mark.parametrize('beta', [2, 1.5])
def test_tv_and_pnorm_match_reference_vectors(golden, beta):
    eng = gpu_engine()
    img = golden['num.img']
    d_img, d_grad = eng.to_device(img), eng.empty(img.shape).zero()
    # tv_norm(img/127.5, beta) alone, scale 1
    res = image_ops.regularizers(eng, d_img, d_grad, MEAN, 1.0, beta, 0.0, 6.0)
    eng.sync()
    assert res.value == pytest.approx(float(golden['num.tv_loss_%g' % beta]), rel=2e-6)
    assert max_rel(d_grad.get(), golden['num.tv_grad_%g' % beta]) < 2e-6
    # p_norm(x/127.5, 6): feed mean = 127.5 so that (img + mean - 127.5)/127.5 == img/127.5
    d_grad.zero()
    res = image_ops.regularizers(eng, d_img, d_grad, np.float32([127.5] * 3), 0.0, 2.0, 1.0, 6.0)
    eng.sync()
    assert res.value == pytest.approx(float(golden['num.p6_loss']), rel=2e-6)
    assert max_rel(d_grad.get(), golden['num.p6_grad']) < 2e-6


def test_regularizers_with_aux_against_oracle():
    eng = gpu_engine()
    rng = np.random.RandomState(4)
    img = rng.uniform(-120, 130, (3, 45, 67)).astype(np.float32)
    aux = rng.uniform(-120, 130, (3, 45, 67)).astype(np.float32)
    g0 = rng.standard_normal(img.shape).astype(np.float32)
    ref = g0.copy()
    ref_loss = regularizer_loss_grad(img, MEAN, ref, 0.7, 5.0, 1.5, 2.0, 6.0, aux, 10.0)
    d_grad = eng.to_device(g0)
    res = image_ops.regularizers(eng, eng.to_device(img), d_grad, MEAN, 0.7 * 5.0, 1.5, 0.7 * 2.0,
                                 6.0, eng.to_device(aux), 0.7 * 10.0)
    eng.sync()
    assert res.value == pytest.approx(ref_loss, rel=1e-5)
    assert max_rel(d_grad.get(), ref) < 1e-5


def test_cut_and_put_tile_are_roll_then_slice():
    eng = gpu_engine()
    rng = np.random.RandomState(5)
    img = rng.standard_normal((3, 37, 52)).astype(np.float32)
    d_img = eng.to_device(img)
    for roll in [(0, 0), (8, -16), (-51, 36), (104, -74)]:
        rolled = num_ops.roll_xy(img.copy(), roll)
        rect = (5, 30, 11, 49)
        tile = eng.empty((3, 25, 38))
        image_ops.cut_tile(eng, d_img, roll, rect, tile)
        assert np.array_equal(tile.get(), rolled[:, 5:30, 11:49])
        # put: writing the tile of a rolled gradient back un-rolls it
        full = eng.empty(img.shape).zero()
        image_ops.put_tile(eng, full, roll, rect, tile)
        expect = np.zeros_like(img)
        expect[:, 5:30, 11:49] = rolled[:, 5:30, 11:49]
        assert np.array_equal(full.get(), num_ops.roll_xy(expect, (-roll[0], -roll[1])))


def _quad(eng, target_dev):
    diff = eng.empty(target_dev.shape)

    def f(x):
        diff.copy_from(x)
        image_ops.axpy(eng, -1.0, target_dev, diff)
        loss = image_ops.dot(eng, diff, diff)
        image_ops.scale(eng, 2.0, diff)
        return loss, diff
    return f


@pytest.mark.parametrize('biased', [0, 1])
def test_adam_matches_reference_trajectory(golden, biased):
    """optimizers.AdamOptimizer incl. roll / un-roll (no-ops on un-rolled device state)."""
    eng = gpu_engine()
    params = eng.to_device(golden['opt.x0'])
    opt = AdamOptimizer(eng, params, step_size=15, bp1=1 - 1 / 20, decay=0.05, power=0.5,
                        biased_g1=bool(biased))
    f = _quad(eng, eng.to_device(golden['opt.target']))
    for i, xy in enumerate(golden['opt.rolls']):
        opt.roll(xy)
        avg, loss = opt.update(f)
        opt.roll(-xy)
        # 5e-6: BLAS saxpy may fuse a*x+y, the kernel rounds the product first
        assert max_rel(avg.get(), golden['opt.adam_biased%d.avg' % biased][i]) < 5e-6
        assert loss == pytest.approx(golden['opt.adam_biased%d.loss' % biased][i], rel=1e-5)
    assert max_rel(params.get(), golden['opt.adam_biased%d.params' % biased]) < 5e-6


def test_lbfgs_matches_reference_trajectory(golden):
    eng = gpu_engine()
    params = eng.to_device(golden['opt.x0'])
    tgt = eng.to_device(golden['opt.target'])
    scale = golden['opt.lbfgs.scale']
    d_scale = eng.to_device(scale)
    work = eng.empty(params.shape)

    def f(x):
        # d = (x - t) * s ; loss = sum d^2 ; grad = 2 d s   (host multiply keeps the test simple)
        d = (x.get() - golden['opt.target']) * scale
        work.set((2 * d * scale).astype(np.float32))
        return float(np.sum(d * d, dtype=np.float64)), work
    opt = LBFGSOptimizer(eng, params)
    for i in range(len(golden['opt.lbfgs.loss'])):
        p, loss = opt.update(f)
        assert max_rel(p.get(), golden['opt.lbfgs.params'][i]) < 5e-4
        assert loss == pytest.approx(golden['opt.lbfgs.loss'][i], rel=5e-3, abs=1e-3)
    del tgt, d_scale


def test_fused_lbfgs_step_changes_no_bit(monkeypatch):
    """The fused passes of the L-BFGS step (stx_vec_axpy_dot_dev, stx_vec_lbfgs_pair, stx_vec_scale2_axpy)
    against one launch per BLAS-1 call: the same iterates bit for bit -- while the memory grows, past
    n_corr pairs (the oldest dropped), and across a pair the curvature test rejects
    (optimizers.py:74-121)."""
    eng = gpu_engine()
    rng = np.random.RandomState(11)
    shape = (3, 37, 53)
    x0 = rng.uniform(-100, 100, shape).astype(np.float32)
    target = rng.uniform(-100, 100, shape).astype(np.float32)
    scale = np.exp(rng.uniform(-1.5, 1.5, shape)).astype(np.float32)
    runs = []
    for fused in ('1', '0'):
        monkeypatch.setenv('STX_LBFGS_FUSED', fused)
        params = eng.to_device(x0)
        work = eng.empty(shape)
        calls = [0]

        def f(x):
            calls[0] += 1
            d = (x.get() - target) * scale
            g = 2 * d * scale
            if calls[0] == 4:          # the same gradient twice: y = 0, s.y = 0 -> the pair is rejected
                g = f.last
            f.last = g
            work.set(g.astype(np.float32))
            return float(np.sum(d * d, dtype=np.float64)), work
        opt = LBFGSOptimizer(eng, params, n_corr=4)
        assert opt.fused == (fused == '1')
        trail = []
        for i in range(12):
            p, loss = opt.update(f)
            trail.append((p.get().copy(), loss, len(opt.sk)))
        runs.append(trail)
    assert [t[2] for t in runs[0]] == [t[2] for t in runs[1]]
    assert [t[2] for t in runs[0]] == [1, 2, 2, 3] + [4] * 8
    for (pa, la, _), (pb, lb, _) in zip(*runs):
        assert np.array_equal(pa, pb) and la == lb


def test_step_stats_and_uint8():
    eng = gpu_engine()
    rng = np.random.RandomState(6)
    avg = rng.uniform(-140, 160, (3, 33, 41)).astype(np.float32)
    old = rng.uniform(-140, 160, (3, 33, 41)).astype(np.float32)
    d_avg, d_old = eng.to_device(avg), eng.to_device(old)
    upd, tv = image_ops.step_stats(eng, d_avg, d_old)
    xd = avg - np.roll(avg, -1, axis=-1)
    yd = avg - np.roll(avg, -1, axis=-2)
    assert upd == pytest.approx(float(np.mean(abs(avg - old))), rel=1e-5)
    assert tv == pytest.approx(float(np.sqrt(np.mean(xd ** 2 + yd ** 2))), rel=1e-5)
    assert np.array_equal(d_old.get(), avg)
    u8 = image_ops.to_u8(eng, d_avg, MEAN)
    ref = np.uint8(np.clip((avg + MEAN)[::-1].transpose(1, 2, 0), 0, 255))
    assert np.array_equal(u8, ref)


@pytest.mark.parametrize('hw,method', [((52, 75), 'lanczos'), ((26, 40), 'lanczos'),
                                       ((52, 75), 'bilinear'), ((37, 80), 'lanczos')])
def test_resample_is_bit_identical_to_pillow(hw, method):
    """num_utils.resize (num_utils.py:90-108) = PIL 'F'-mode resize per channel."""
    from PIL import Image
    from style_transfer_amd.resample import resample_device
    eng = gpu_engine()
    rng = np.random.RandomState(0)
    a = rng.uniform(-100, 100, (3, 37, 53)).astype(np.float32)
    pil_method = Image.LANCZOS if method == 'lanczos' else Image.BILINEAR
    ref = np.stack([np.asarray(Image.fromarray(a[c]).resize((hw[1], hw[0]), pil_method))
                    for c in range(3)])
    got = resample_device(eng, eng.to_device(a), hw, method).get()
    assert np.array_equal(got, ref)
    clamped = resample_device(eng, eng.to_device(a), hw, method, clamp_min_zero=True).get()
    assert np.array_equal(clamped, np.maximum(0, ref))


@pytest.mark.parametrize('shape,roll,power', [((3, 16, 16), (0, 0), 2), ((3, 37, 53), (8, -16), 2),
                                              ((3, 64, 20), (-24, 40), 1), ((3, 50, 44), (16, 8), 1.5)])
def test_swt_haar_term_against_oracle(shape, roll, power):
    """stx_image_swt_haar vs the oracle's restatement of num_utils.swt_norm(x, 'haar', 1, p) on the
    picture rolled by the iteration's shift (PyWavelets is absent: the oracle itself is held to a
    band-by-band transform in tests/test_oracle_swt.py; no reference vectors exist)."""
    from oracle import num_ops
    eng = gpu_engine('vgg19')
    rng = np.random.RandomState(shape[1])
    img = rng.uniform(-120, 130, shape).astype(np.float32)
    g0 = rng.standard_normal(shape).astype(np.float32)
    scale = 0.37
    rolled = np.roll(img, (roll[1], roll[0]), (1, 2))          # roll = (x, y)
    loss, grad = num_ops.swt_norm_haar1(rolled / np.float32(127.5), power)
    want = g0 + np.float32(scale) * np.roll(grad, (-roll[1], -roll[0]), (1, 2))
    d_img, d_grad = eng.to_device(img), eng.to_device(g0)
    out = image_ops.swt_haar(eng, d_img, d_grad, scale, power, roll=roll)
    eng.sync()
    assert out.value == pytest.approx(scale * loss, rel=2e-5)
    assert np.abs(d_grad.get() - want).max() <= 2e-5 * np.abs(want).max()


def test_cut_and_put_tile_sixteen_byte_path_equals_roll_then_slice():
    """Widths, tile origin and shift multiples of four floats take the 16-byte kernel
    (tile_move4_kernel); anything else the dword one.  Both are roll2 + slice
    (style_transfer.py:619-643, num_utils.py:136-140), bit for bit, wrap included."""
    eng = gpu_engine()
    rng = np.random.RandomState(15)
    img = rng.standard_normal((3, 40, 96)).astype(np.float32)
    d_img = eng.to_device(img)
    for roll in [(0, 0), (8, -16), (-52, 36), (200, -74), (6, 3)]:       # the last one: dword path
        for rect in [(0, 24, 0, 48), (8, 40, 48, 96), (4, 36, 32, 96)]:
            th, tw = rect[1] - rect[0], rect[3] - rect[2]
            rolled = num_ops.roll_xy(img.copy(), roll)
            tile = eng.empty((3, th, tw))
            image_ops.cut_tile(eng, d_img, roll, rect, tile)
            assert np.array_equal(tile.get(), rolled[:, rect[0]:rect[1], rect[2]:rect[3]]), (roll, rect)
            full = eng.empty(img.shape).zero()
            image_ops.put_tile(eng, full, roll, rect, tile)
            expect = np.zeros_like(img)
            expect[:, rect[0]:rect[1], rect[2]:rect[3]] = rolled[:, rect[0]:rect[1], rect[2]:rect[3]]
            assert np.array_equal(full.get(), num_ops.roll_xy(expect, (-roll[0], -roll[1]))), (roll, rect)
            tile.free()
            full.free()


def test_fences_publish_exactly_what_they_closed():
    """stx_fence / stx_fence_wait (the run-ahead step loop): values queued before a fence are
    published by waiting for that fence -- or earlier, when a second fence has to reuse their set
    -- and never depend on work queued after it; asynchronous step statistics equal the
    synchronous ones."""
    eng = gpu_engine()
    rng = np.random.RandomState(16)
    imgs = [rng.uniform(-100, 100, (3, 32, 48)).astype(np.float32) for _ in range(3)]
    d = [eng.to_device(a) for a in imgs]
    grads = [eng.empty(a.shape).zero() for a in imgs]
    want = []
    for a, g in zip(d, grads):
        r = image_ops.regularizers(eng, a, g, MEAN, 5.0, 2.0, 2.0, 6.0)
        eng.sync()
        want.append(r.value)
        g.zero()
    eng.sync()
    # three values behind three fences; the third fence reuses the first one's set
    pending, tickets = [], []
    for a, g in zip(d, grads):
        pending.append(image_ops.regularizers(eng, a, g, MEAN, 5.0, 2.0, 2.0, 6.0))
        tickets.append(eng.fence())
    assert pending[0].value == want[0]                 # published when its set was needed again
    assert np.isnan(pending[2].value)
    eng.wait_fence(tickets[0])                          # (an old ticket: nothing to do)
    eng.wait_fence(tickets[2])
    assert pending[2].value == want[2]
    eng.wait_fence(tickets[1])
    assert pending[1].value == want[1]
    # statistics: the asynchronous form against the synchronous one
    avg = rng.uniform(-140, 160, (3, 33, 41)).astype(np.float32)
    old = rng.uniform(-140, 160, (3, 33, 41)).astype(np.float32)
    d_avg, d_old, d_old2 = eng.to_device(avg), eng.to_device(old), eng.to_device(old)
    sync_stats = image_ops.step_stats(eng, d_avg, d_old)
    lazy = image_ops.step_stats_async(eng, d_avg, d_old2)
    eng.wait_fence(eng.fence())
    assert lazy.values() == sync_stats
    assert np.array_equal(d_old2.get(), avg)
