"""Helpers for the -m gpu parity tests (HIP path through the C ABI vs the numpy oracle)."""

import numpy as np
import pytest

from oracle.caffe_net import synthetic_weights
from style_transfer_amd.netspec import builtin_net

_ENGINES = {}


def gpu_engine(model='vgg19', seed=0):
    """A cached TileEngine with the same seeded synthetic weights the oracle uses."""
    from style_transfer_amd import lib
    from style_transfer_amd.engine import TileEngine
    if lib.device_count() < 1:
        pytest.fail('no GPU visible: -m gpu tests need an MI355X and the built libstx.so')
    key = (model, seed)
    if key not in _ENGINES:
        net = builtin_net(model)
        _ENGINES[key] = TileEngine(net, 0, synthetic_weights(net.as_dicts(), seed))
    return _ENGINES[key]


def max_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
