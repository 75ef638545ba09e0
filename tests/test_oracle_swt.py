"""The SWT regularizer of the reference (num_utils.py:179-196) for its command-line defaults
('haar', one level).  PyWavelets is absent here and upstream-unpinned, so there are no reference
vectors (parity unpinned, oracle/num_ops.py); what can be held down is the algebra: the closed form
used by the oracle and the GPU equals a band-by-band stationary Haar transform with the low-low band
zeroed, and the transform with nothing zeroed is the identity."""
import numpy as np

from oracle import num_ops


def test_closed_form_equals_filter_bank():
    rng = np.random.RandomState(0)
    for n in (8, 16, 64):
        ch = rng.standard_normal((n, n))
        want = num_ops.swt_haar1_filterbank(ch)
        got = num_ops.swt_haar1_detail(ch[None].astype(np.float32))[0]      # n is a power of two: no padding
        assert np.abs(got - want).max() < 2e-6


def test_filter_bank_is_a_perfect_reconstruction():
    rng = np.random.RandomState(1)
    ch = rng.standard_normal((16, 16))
    s = np.sqrt(0.5)
    lo, hi = s * (ch + np.roll(ch, -1, 0)), s * (ch - np.roll(ch, -1, 0))
    back = 0.5 * (s * (lo + hi) + s * (np.roll(lo, 1, 0) - np.roll(hi, 1, 0)))
    assert np.abs(back - ch).max() < 1e-12
    # constants live entirely in the low-low band
    assert np.abs(num_ops.swt_haar1_filterbank(np.full((8, 8), 3.0))).max() < 1e-12


def test_padding_and_crop_follow_the_reference():
    # 5 x 11 plane: padded to 16 x 16, 5 / 6 rows and 2 / 3 columns (num_utils.py:165-176)
    assert num_ops._pad_width((3, 5, 11), (1, 16, 16)) == [(0, 0), (5, 6), (2, 3)]
    x = np.random.RandomState(2).standard_normal((3, 5, 11)).astype(np.float32)
    d = num_ops.swt_haar1_detail(x)
    assert d.shape == x.shape
    loss, grad = num_ops.swt_norm_haar1(x, 2)
    assert np.isclose(loss, float((d.astype(np.float64) ** 2).sum()), rtol=1e-5)
    assert np.allclose(grad, 2 * d)
