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


MEAN_BGR = np.float32([103.939, 116.779, 123.68])


def raw_to_u8(raw):
    """The reference's picture of a mean-subtracted BGR array (style_transfer.py: deprocess): add the
    mean, BGR -> RGB, clip, truncate.  Pinned on the fixture's own (final_raw, final_u8) pair by
    tests/test_oracle_golden.py."""
    x = raw + MEAN_BGR[:, None, None]
    return np.uint8(np.clip(x[::-1].transpose(1, 2, 0), 0, 255))


def cfg4_reference_branches(golden):
    """The reference's own trajectories on the config-4 miniature (make_golden.py section 4c): the
    committed run, and the runs of the same reference code with its Convolution layer computed by other
    float32 implementations (tests/golden/branch_sets.py cfg4 -> cfg4_branches.npz: torch's conv2d, per-tap
    and K-blocked SGEMMs in several orders, and noise of the amplitude those implementations differ by).  The
    trajectory branches at ReLU / max-pooling near-ties of its 30 x 33-pixel tiles: 4 of the 14
    implementations (torch's among them) and 1 of 8 noise runs leave the committed one by 5.7e-4 at the
    second step.  [{log [5][4], final_raw, final_u8, runs}], the committed run first."""
    import os
    out = [dict(log=np.float64(golden['e2e_cfg4.log']), final_raw=golden['e2e_cfg4.final_raw'],
                final_u8=golden['e2e_cfg4.final_u8'], runs=0)]
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cfg4_branches.npz'))
    for log, raw, runs in zip(d['logs'], d['final_raw'], d['runs']):
        if np.allclose(log[:, 2], out[0]['log'][:, 2], rtol=2e-4):
            out[0]['runs'] += int(runs)
        else:
            out.append(dict(log=log, final_raw=raw, final_u8=raw_to_u8(raw), runs=int(runs)))
    return out


def lbfgs_reference_outcomes(golden):
    """The reference's own outcomes on the e2e_lbfgs fixture (make_golden.py section 4b): the committed
    run, and the distinct final pictures the same reference code lands on with its Convolution layer or
    its Gram matrix computed by other float32 implementations / under noise of their amplitude
    (tests/golden/branch_sets.py lbfgs -> lbfgs_branches.npz).  [{log [5][4], final_raw, runs}], the
    committed run first."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lbfgs_branches.npz'))
    out = [dict(log=np.float64(log), final_raw=raw, runs=int(runs))
           for log, raw, runs in zip(d['logs'], d['final_raw'], d['runs'])]
    assert np.array_equal(out[0]['final_raw'], golden['e2e_lbfgs.final_raw'])
    return out


def matching_branch(branches, losses, rtol=2e-4):
    """The reference trajectory whose losses `losses` follows step by step to rtol (None: none of them)."""
    for b in branches:
        if len(losses) == len(b['log']) and np.allclose(losses, b['log'][:, 2], rtol=rtol):
            return b
    return None
