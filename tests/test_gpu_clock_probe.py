"""stx_clock_marks: the shader clock read inside an engine's stream (bench.py's roofline.clock_mhz).
Two one-wave kernels of 20 microseconds per tile evaluation; they must leave the results alone and
report a clock inside the part's range."""

import numpy as np
import pytest

from tests.gpu_helpers import gpu_engine
from tests.helpers import DEFAULT_STYLE_LAYERS, normalized_weights

pytestmark = pytest.mark.gpu


def test_clock_marks_record_two_readings_per_tile_and_change_nothing():
    eng = gpu_engine('vgg19')
    rng = np.random.RandomState(1)
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    th = tw = 256
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
    assert len(mhz) == 6
    assert np.all(mhz > 300) and np.all(mhz < 3000), mhz
    assert eng.clock_marks_read() == []                     # (reading clears)
    eng.clock_marks(False)
    eng.sc_grad_tile(tile, (0, 0), (0, 0), cl, sl, {}, cw, sw)
    assert eng.clock_marks_read() == []
