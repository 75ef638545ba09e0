"""stx_clock_marks: the shader clock read INSIDE the dominant kernel (bench.py's roofline.clock_mhz).
One workgroup of every 2-D Winograd convolution launch times its chunk loop with the core-cycle
counter and the constant 100 MHz counter; the marks must leave the results alone and report a clock
inside the part's range."""

import numpy as np
import pytest

from tests.gpu_helpers import gpu_engine
from tests.helpers import DEFAULT_STYLE_LAYERS, normalized_weights

pytestmark = pytest.mark.gpu


def test_clock_marks_record_one_reading_per_winograd_launch_and_change_nothing():
    eng = gpu_engine('vgg19')
    rng = np.random.RandomState(1)
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    th = tw = 512
    contents = [{'conv4_2': np.abs(rng.standard_normal(eng.feature_shape('conv4_2', th, tw))).astype(np.float32)}]
    styles = [{l: np.tril(0.05 * rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32)
               for l in sl}]
    eng.set_contents_and_styles(contents, styles)
    tile = rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)
    want_loss, want_grad = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
    assert eng.clock_marks_read() == []
    eng.clock_marks(True)
    for _ in range(3):
        loss, grad = eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
        assert loss == want_loss and np.array_equal(grad, want_grad)
    mhz = np.array(eng.clock_marks_read())
    # conv1_2 .. conv5_1 forward (12 layers) and backward (12), three evaluations (fewer where
    # the shape rule hands a plane to the four-wave kernel, which carries no mark)
    assert len(mhz) % 3 == 0 and 3 * 12 <= len(mhz) <= 3 * 24, len(mhz)
    real = mhz[mhz > 0]                                     # (0: a chunk loop shorter than a microsecond)
    assert len(real) >= 3 * 12
    assert np.all(real > 300) and np.all(real < 3000), mhz
    assert eng.clock_marks_read() == []                     # (reading clears)
    eng.clock_marks(False)
    eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
    assert eng.clock_marks_read() == []
    # the per-group profile carries the same marks
    eng.profile(True)
    eng.clock_marks(True)
    eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
    rows = eng.profile_read(clock=True)
    assert any(r[0] == 'fwd conv1_2' and 300 < r[3] < 3000 for r in rows), rows[:8]
    assert all(r[3] == 0 for r in rows if r[0].startswith('gram') or r[0].startswith('symm'))
    eng.profile(False)
    eng.clock_marks(False)
