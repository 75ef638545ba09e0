#!/usr/bin/env python3
"""What do two float32 implementations of Caffe's Convolution layer differ by -- measured, not assumed.

The branch sets of the chaotic L-BFGS fixtures (cfg4_sensitivity.py, lbfgs_sensitivity.py) are made by
running the reference's own Python with its convolution outputs perturbed.  Round 5 sized that
perturbation from the error of the GPU kernels against float64 (+-3e-7 of a blob's maximum per element).
This script sizes it from the REFERENCE'S SIDE instead: the same convolution (im2col + SGEMM semantics,
float32 storage, float32 accumulation) computed by several legitimate float32 implementations that differ
only in the order / blocking of their sums --

  sgemm      oracle.layers.conv_forward: one im2col SGEMM over K = 9 Cin (OpenBLAS, threaded) -- the
             implementation the committed fixtures were made with
  torch      torch.nn.functional.conv2d on the CPU in float32 (oneDNN)
  taps       nine SGEMMs of K = Cin, one per filter tap, added in float32 tap by tap (ky outer)
  taps_rev   the same with the taps in reverse order
  chunk16    K split into blocks of 16 input channels x 9 taps, partial products added in float32
  chunk64    blocks of 64 input channels
  pairwise   chunk16's partial products added as a balanced tree

-- on the convolution inputs the fixture's own forward passes see.  Per call: max and rms of
|y_a - y_b| / max |y| over every pair, and of each against a float64 convolution.

    python tests/golden/fp32_noise.py [cfg4|lbfgs|stable]

Build container only (imports /root/reference through make_golden).  Writes
tests/golden/fp32_noise_calibration.json; prints the table DESIGN section 4 quotes.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from oracle import layers as L  # noqa: E402

PLAIN = L.conv_forward


def _cols(x, pad=1):
    cin, h, w = x.shape
    xp = np.zeros((cin, h + 2 * pad, w + 2 * pad), np.float32)
    xp[:, pad:pad + h, pad:pad + w] = x
    return xp


def conv_taps(x, w, b, pad=1, order=None):
    cin, h, wd = x.shape
    cout = w.shape[0]
    xp = _cols(x, pad)
    y = np.zeros((cout, h * wd), np.float32)
    for t in (order or range(9)):
        ky, kx = divmod(t, 3)
        y += w[:, :, ky, kx] @ np.ascontiguousarray(xp[:, ky:ky + h, kx:kx + wd]).reshape(cin, -1)
    y += b.astype(np.float32)[:, None]
    return y.reshape(cout, h, wd)


def conv_taps_rev(x, w, b, pad=1):
    return conv_taps(x, w, b, pad, order=range(8, -1, -1))


def _partials(x, w, block, pad=1):
    cin, h, wd = x.shape
    cout = w.shape[0]
    xp = _cols(x, pad)
    parts = []
    for c0 in range(0, cin, block):
        c1 = min(cin, c0 + block)
        cols = np.empty((c1 - c0, 3, 3, h, wd), np.float32)
        for ky in range(3):
            for kx in range(3):
                cols[:, ky, kx] = xp[c0:c1, ky:ky + h, kx:kx + wd]
        parts.append(w[:, c0:c1].reshape(cout, -1) @ cols.reshape(-1, h * wd))
    return parts


def conv_chunk(x, w, b, pad=1, block=16):
    parts = _partials(x, w, block, pad)
    y = parts[0].copy()
    for p in parts[1:]:
        y += p
    y += b.astype(np.float32)[:, None]
    return y.reshape(w.shape[0], x.shape[1], x.shape[2])


def conv_chunk64(x, w, b, pad=1):
    return conv_chunk(x, w, b, pad, 64)


def conv_pairwise(x, w, b, pad=1):
    parts = _partials(x, w, 16, pad)
    while len(parts) > 1:
        nxt = [parts[i] + parts[i + 1] for i in range(0, len(parts) - 1, 2)]
        if len(parts) % 2:
            nxt.append(parts[-1])
        parts = nxt
    y = parts[0] + b.astype(np.float32)[:, None]
    return y.reshape(w.shape[0], x.shape[1], x.shape[2])


def conv_torch(x, w, b, pad=1):
    import torch
    with torch.no_grad():
        y = torch.nn.functional.conv2d(torch.from_numpy(np.ascontiguousarray(x))[None], torch.from_numpy(w),
                                       torch.from_numpy(b.astype(np.float32)), padding=pad)
    return y[0].numpy()


VARIANTS = {'sgemm': PLAIN, 'torch': conv_torch, 'taps': conv_taps, 'taps_rev': conv_taps_rev,
            'chunk16': conv_chunk, 'chunk64': conv_chunk64, 'pairwise': conv_pairwise}


def _f64(x, w, b, pad):
    cin, h, wd = x.shape
    cout = w.shape[0]
    xp = np.zeros((cin, h + 2 * pad, wd + 2 * pad), np.float64)
    xp[:, pad:pad + h, pad:pad + wd] = x
    cols = np.empty((cin, 3, 3, h, wd), np.float64)
    for ky in range(3):
        for kx in range(3):
            cols[:, ky, kx] = xp[:, ky:ky + h, kx:kx + wd]
    y = w.reshape(cout, -1).astype(np.float64) @ cols.reshape(-1, h * wd) + b.astype(np.float64)[:, None]
    return y.reshape(cout, h, wd)


class Recorder:
    """Stands in for oracle.layers.conv_forward during one run of a fixture: returns the plain result
    and keeps, per (Cin, Cout) of the call, the differences between the variants on that call's data."""

    def __init__(self, max_calls=400):
        self.rows, self.calls, self.max_calls = {}, 0, max_calls

    def __call__(self, x, w, b, pad=1):
        y = PLAIN(x, w, b, pad)
        self.calls += 1
        if self.calls <= self.max_calls and w.shape[-1] == 3:
            ys = {k: f(x, w, b, pad) for k, f in VARIANTS.items() if k != 'sgemm'}
            ys['sgemm'] = y
            ref = _f64(x, w, b, pad)
            m = float(np.abs(ref).max()) or 1.0
            names = sorted(ys)
            row = self.rows.setdefault((w.shape[1], w.shape[0]), {'pair_max': [], 'pair_rms': [], 'f64_max': [],
                                                                  'f64_rms': [], 'pixels': []})
            row['pixels'].append(x.shape[1] * x.shape[2])
            for i, a in enumerate(names):
                d = (ys[a].astype(np.float64) - ref) / m
                row['f64_max'].append(float(np.abs(d).max()))
                row['f64_rms'].append(float(np.sqrt((d * d).mean())))
                for c in names[i + 1:]:
                    d = (ys[a].astype(np.float64) - ys[c]) / m
                    row['pair_max'].append(float(np.abs(d).max()))
                    row['pair_rms'].append(float(np.sqrt((d * d).mean())))
        return y


def summarize(rows):
    out = {}
    print('%-12s %6s %8s | pairs of float32 implementations: max |ya - yb| / max|y|  rms | each against float64: max  rms'
          % ('Cin -> Cout', 'calls', 'pixels'))
    for (cin, cout), r in sorted(rows.items()):
        out['%d->%d' % (cin, cout)] = {k: [float(np.median(v)), float(np.max(v))] for k, v in r.items()}
        print('%4d -> %4d %6d %8d |  median %.2e  largest %.2e   %.2e | %.2e  %.2e'
              % (cin, cout, len(r['pixels']), int(np.median(r['pixels'])), np.median(r['pair_max']),
                 np.max(r['pair_max']), np.median(r['pair_rms']), np.median(r['f64_max']), np.median(r['f64_rms'])))
    allmax = np.concatenate([r['pair_max'] for r in rows.values()])
    allrms = np.concatenate([r['pair_rms'] for r in rows.values()])
    out['all'] = {'pair_max_median': float(np.median(allmax)), 'pair_max_p90': float(np.percentile(allmax, 90)),
                  'pair_max_largest': float(allmax.max()), 'pair_rms_median': float(np.median(allrms))}
    print('all calls: pair max: median %.2e, 90th percentile %.2e, largest %.2e; pair rms median %.2e'
          % (out['all']['pair_max_median'], out['all']['pair_max_p90'], out['all']['pair_max_largest'],
             out['all']['pair_rms_median']))
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
    mg.install_stubs()
    sys.argv = ['style_transfer.py', '-ci', 'c.png', '-si', 's.png']
    import config_system
    import num_utils
    import style_transfer as st
    import fixtures_lbfgs as fx
    rec = Recorder()
    with contextlib.redirect_stdout(io.StringIO()):
        fx.run_fixture(st, config_system, num_utils, which, conv=rec)
    res = summarize(rec.rows)
    path = os.path.join(HERE, 'fp32_noise_calibration.json')
    old = json.load(open(path)) if os.path.exists(path) else {}
    old[which] = res
    json.dump(old, open(path, 'w'), indent=1, sort_keys=True)
    num_utils.POOL.shutdown()


if __name__ == '__main__':
    main()
