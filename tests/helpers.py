"""Shared helpers for the parity tests (oracle side)."""

import numpy as np

from oracle.caffe_net import synthetic_weights
from oracle.tile_path import OracleModel
from style_transfer_amd.netspec import builtin_net

MEAN = np.float32((103.939, 116.779, 123.68)).reshape(3, 1, 1)
DEFAULT_STYLE_LAYERS = ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']


def u8_to_params(u8):
    """RGB HWC uint8 -> BGR CHW float32 minus mean (style_transfer.py:388-393)."""
    return np.ascontiguousarray(np.float32(u8).transpose(2, 0, 1)[::-1] - MEAN)


def normalized_weights(names, master):
    """``StyleTransfer.parse_weights`` for unweighted names (style_transfer.py:684-698)."""
    return list(names), {n: master / len(names) for n in names}


def make_oracle(model_name, seed=0):
    net = builtin_net(model_name)
    layers = net.as_dicts()
    return OracleModel(layers, synthetic_weights(layers, seed)), net


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
