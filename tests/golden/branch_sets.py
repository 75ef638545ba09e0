#!/usr/bin/env python3
"""Which trajectories does the REFERENCE'S OWN L-BFGS run take when its float32 kernels are another
float32 implementation of the same arithmetic?  (Build container only: imports /root/reference.)

Replaces round 5's cfg4_sensitivity.py / lbfgs_sensitivity.py, whose perturbation (uniform +-3e-7 of a
blob's maximum per element) was sized from the GPU kernels' error.  Here the perturbation is what the
reference's side itself allows:

  * REAL implementations -- the reference's Python exactly as make_golden.py runs it, with the Convolution
    layer behind its pycaffe calls computed by each of fp32_noise.VARIANTS and their relatives (torch-CPU
    conv2d; per-tap SGEMMs in several tap orders; K blocked by 8 .. 128 input channels, added in sequence,
    in reverse and pairwise): float32 storage, float32 products and sums, only the order of the sums
    differs -- and, for the fixture whose Gram matrices matter (AVE pooling: no max-pooling ties), the
    reference's gram_matrix computed with the pixel axis blocked (float32 partial Grams added in sequence);
  * SYNTHETIC noise at the amplitude those implementations measure (fp32_noise.py: pairs of them differ
    by 3.6e-7 of max |y| at the largest element, median over calls, 8.3e-7 at most, rms 6e-8): every
    convolution output moved by a normal deviate of rms 6e-8 max |y|, clipped at 8e-7 max |y|.

    python tests/golden/branch_sets.py cfg4|lbfgs|stable [--noise N] [--procs P]

Writes tests/golden/<fixture>_runs.json (every run: name, per-step loss difference to the unperturbed run,
final-picture distance) and, for cfg4 / lbfgs, tests/golden/<fixture>_branches.npz (one representative log
and final picture per distinct outcome, the committed fixture first: what tests/helpers.py reads).
For `stable` it reports whether every run stays inside the 2e-4 band the GPU test holds it to.
"""
import argparse
import json
import os
import sys

# (before numpy loads OpenBLAS: the worker processes, started with `spawn`, inherit it -- P processes x T
# threads should not exceed the cores)
os.environ.setdefault('OPENBLAS_NUM_THREADS', '2')
os.environ.setdefault('OMP_NUM_THREADS', '2')

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

NOISE_RMS, NOISE_CLIP = 6e-8, 8e-7


def conv_variants():
    import fp32_noise as fn
    v = {'torch': fn.conv_torch, 'taps': fn.conv_taps, 'taps_rev': fn.conv_taps_rev, 'pairwise16': fn.conv_pairwise}
    for seed in (1, 2, 3):
        order = list(np.random.RandomState(seed).permutation(9))
        v['taps_perm%d' % seed] = lambda x, w, b, pad=1, o=order: fn.conv_taps(x, w, b, pad, order=o)
    for blk in (8, 16, 32, 64, 128):
        v['chunk%d' % blk] = lambda x, w, b, pad=1, k=blk: fn.conv_chunk(x, w, b, pad, k)

    def chunk_rev(x, w, b, pad=1):
        parts = fn._partials(x, w, 16, pad)[::-1]
        y = parts[0].copy()
        for p in parts[1:]:
            y += p
        y += b.astype(np.float32)[:, None]
        return y.reshape(w.shape[0], x.shape[1], x.shape[2])
    v['chunk16_rev'] = chunk_rev

    def pairwise8(x, w, b, pad=1):
        parts = fn._partials(x, w, 8, pad)
        while len(parts) > 1:
            nxt = [parts[i] + parts[i + 1] for i in range(0, len(parts) - 1, 2)]
            if len(parts) % 2:
                nxt.append(parts[-1])
            parts = nxt
        y = parts[0] + b.astype(np.float32)[:, None]
        return y.reshape(w.shape[0], x.shape[1], x.shape[2])
    v['pairwise8'] = pairwise8
    return v


def gram_variants():
    """num_utils.gram_matrix (num_utils.py:53-59: ssyrk over all pixels / size, lower triangle) with the pixel
    axis blocked: float32 partial products added in float32."""
    def blocked(block, reverse=False):
        def gram(feat):
            f = feat.reshape(feat.shape[0], -1)
            starts = list(range(0, f.shape[1], block))
            if reverse:
                starts = starts[::-1]
            g = np.zeros((f.shape[0], f.shape[0]), np.float32)
            for s in starts:
                p = np.ascontiguousarray(f[:, s:s + block])
                g += p @ p.T
            return np.tril(g * np.float32(1 / f.size))          # ssyrk's alpha = 1 / size, lower triangle
        return gram
    return {'gram_blk64': blocked(64), 'gram_blk256': blocked(256), 'gram_blk1024': blocked(1024),
            'gram_blk256_rev': blocked(256, True), 'gram_blk100': blocked(100), 'gram_blk37': blocked(37)}


def noisy_conv(seed):
    import fp32_noise as fn
    rng = np.random.RandomState(seed)

    def conv(x, w, b, pad=1):
        y = fn.PLAIN(x, w, b, pad)
        e = np.clip(rng.standard_normal(y.shape) * NOISE_RMS, -NOISE_CLIP, NOISE_CLIP)
        return (y + e * np.abs(y).max()).astype(np.float32)
    return conv


_STATE = {}


def _init():
    import torch
    torch.set_num_threads(int(os.environ.get('OPENBLAS_NUM_THREADS', '2')))
    import make_golden as mg
    mg.install_stubs()
    sys.argv = ['style_transfer.py', '-ci', 'c.png', '-si', 's.png']
    import config_system
    import num_utils
    import style_transfer as st
    import fixtures_lbfgs as fx
    _STATE.update(cs=config_system, nu=num_utils, st=st, fx=fx)


def _run(job):
    which, name = job
    if not _STATE:
        _init()
    conv = gram = None
    if name.startswith('noise'):
        conv = noisy_conv(1000 + int(name[5:]))
    elif name.startswith('gram_'):
        gram = gram_variants()[name]
    elif name != 'as_is':
        conv = conv_variants()[name]
    log, raw = _STATE['fx'].run_fixture(_STATE['st'], _STATE['cs'], _STATE['nu'], which, conv=conv, gram=gram)[:2]
    return name, log, np.float32(raw)


def group(base_log, base_img, runs, by_picture):
    """One representative per distinct outcome.  by_picture: two runs are the same outcome when their final
    pictures agree to 0.05 (the AVE fixture: no trajectory split, a set of pictures); else when their losses
    agree to the tests' 2e-4 at every step (the MAX fixtures: the trajectory itself splits)."""
    reps = [[np.float64(base_log), np.float32(base_img), 1, ['committed']]]
    for name, log, img in runs:
        for r in reps:
            same = np.abs(img - r[1]).max() < 0.05 if by_picture else np.allclose(log[:, 2], r[0][:, 2], rtol=2e-4)
            if same:
                r[2] += 1
                r[3].append(name)
                break
        else:
            reps.append([np.float64(log), np.float32(img), 1, [name]])
    return reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fixture', choices=['cfg4', 'lbfgs', 'stable'])
    ap.add_argument('--noise', type=int, default=8)
    ap.add_argument('--procs', type=int, default=4)
    ap.add_argument('--only', nargs='*')
    a = ap.parse_args()
    names = ['as_is'] + sorted(conv_variants())
    if a.fixture == 'lbfgs':
        names += sorted(gram_variants())
    names += ['noise%d' % k for k in range(a.noise)]
    if a.only:
        names = a.only
    import multiprocessing as mp
    with mp.get_context('spawn').Pool(a.procs, initializer=_init) as pool:
        results = []
        for name, log, raw in pool.imap(_run, [(a.fixture, n) for n in names]):
            results.append((name, log, raw))
            print('%-14s done: losses %s' % (name, ' '.join('%.7e' % v for v in log[:, 2])), flush=True)
    base = next(r for r in results if r[0] == 'as_is') if not a.only else results[0]
    golden = np.load(os.path.join(HERE, 'reference_vectors.npz'))
    key = {'cfg4': 'e2e_cfg4', 'lbfgs': 'e2e_lbfgs', 'stable': 'e2e_stable'}[a.fixture]
    have = key + '.log' in golden
    ref_log = golden[key + '.log'] if have else base[1]
    ref_img = golden[key + '.final_raw'] if have else base[2]
    rows = []
    for name, log, raw in results:
        d = np.abs(raw - ref_img)
        rows.append({'run': name, 'loss_rel': [float(v) for v in np.abs(log[:, 2] / ref_log[:, 2] - 1)],
                     'update_rel': [float(v) for v in np.abs(log[:, 1] / ref_log[:, 1] - 1)],
                     'picture_max': float(d.max()), 'picture_mean': float(d.mean())})
        print('%-14s loss rel. diff per step %s   final picture |diff| max %.3f mean %.5f'
              % (name, ' '.join('%.1e' % v for v in rows[-1]['loss_rel']), d.max(), d.mean()))
    json.dump({'against': 'committed fixture' if have else 'this script\'s as_is run', 'runs': rows},
              open(os.path.join(HERE, a.fixture + '_runs.json'), 'w'), indent=1)
    worst = max(max(r['loss_rel']) for r in rows)
    print('largest loss difference to the %s over %d runs: %.2e' % ('committed fixture' if have else 'as_is run',
                                                                      len(rows), worst))
    if a.fixture != 'stable' and not a.only:
        reps = group(ref_log, ref_img, results, by_picture=a.fixture == 'lbfgs')
        np.savez_compressed(os.path.join(HERE, a.fixture + '_branches.npz'),
                            logs=np.float64([r[0] for r in reps]), final_raw=np.float32([r[1] for r in reps]),
                            runs=np.int64([r[2] for r in reps]), names=np.array([' '.join(r[3]) for r in reps]))
        for i, r in enumerate(reps):
            d = np.abs(r[1] - reps[0][1])
            print('outcome %d: %2d runs (%s): loss against the committed run %s, picture max %.3f mean %.4f'
                  % (i, r[2], ' '.join(r[3]), ' '.join('%.1e' % v for v in np.abs(r[0][:, 2] / reps[0][0][:, 2] - 1)),
                     d.max(), d.mean()))


if __name__ == '__main__':
    main()
