"""The whole multi-scale schedule on the GPU against the run the reference's own
``transfer_multiscale`` produced (tests/golden: 2 scales, 2x2 tiles, 3+2 Adam steps, seed 5).

Per-step losses must agree to 1e-4 relative.  The final averaged image is compared per pixel
with an absolute tolerance of 0.5 (on a 0..255 scale): five normalised-gradient steps amplify
the few max-pool / ReLU decision flips discussed in test_gpu_tile_path.py."""

import numpy as np
import pytest
from PIL import Image

from style_transfer_amd.config_system import parse_args
from style_transfer_amd.farm import TileFarm
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.transfer import StyleTransfer
from style_transfer_amd.weights import synthetic_weights

pytestmark = pytest.mark.gpu


class _BehindCallback:
    """A callback the step loop may serve one iteration late (it declares which iterations' images
    it reads: the second one here, which therefore is collected before the third is queued)."""

    def __init__(self, log):
        self.log, self.asked = log, []

    def wants_image(self, n):
        self.asked.append(n)
        return n == 2

    def __call__(self, **kw):
        self.log.append((kw['step'], kw['update_size'], kw['loss'], kw['tv_loss']))


@pytest.mark.parametrize('run_ahead', [False, True])
def test_transfer_multiscale_matches_reference_run(golden, run_ahead):
    """run_ahead: the host queues iteration i + 1 before it collects iteration i (a callback with
    ``wants_image``: the command line's Progress) -- same values, same order as the reference's
    blocking loop."""
    from argparse import Namespace
    argv = str(golden['e2e.argv']).split()
    state = Namespace()
    args = parse_args(state, argv, config_py=False)
    net = builtin_net('vgg19')
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    st = StyleTransfer(farm, args, state)
    log = []
    np.random.seed(args.seed)
    behind = _BehindCallback(log)
    st.transfer_multiscale([Image.fromarray(golden['e2e.content_u8'])],
                           [Image.fromarray(golden['e2e.style_u8'])],
                           callback=behind if run_ahead else lambda **kw: log.append(
                               (kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
    assert behind.asked == ([1, 2, 4] if run_ahead else [])      # (the last iteration of a scale is always collected)
    ref = golden['e2e.log']
    got = np.float64(log)
    assert got.shape == ref.shape
    assert np.array_equal(got[:, 0], ref[:, 0])
    assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-4), (got[:, 2], ref[:, 2])     # loss
    assert np.allclose(got[:, 1], ref[:, 1], rtol=1e-3)                             # update size
    assert np.allclose(got[:, 3], ref[:, 3], rtol=1e-3)                             # tv statistic
    final = st.current_raw.get()
    assert final.shape == golden['e2e.final_raw'].shape
    assert np.abs(final - golden['e2e.final_raw']).max() < 0.5
    u8 = np.asarray(st.current_output)
    assert np.abs(u8.astype(int) - golden['e2e.final_u8'].astype(int)).max() <= 1
    assert farm.tile_evals == 4 * 3 + 4 * 2
    from style_transfer_amd import lib
    assert sum(e.query(lib.Q_TILE_EVALS) for e in farm.engines) == farm.tile_evals
    farm.close()


def test_run_ahead_step_loop_changes_no_bit(golden):
    """The step loop running one iteration ahead of the GPU (fences, asynchronous statistics)
    against the blocking schedule (every iteration collected before the next is queued): the same
    losses, statistics and final image, bit for bit, and the same global RNG state at the end."""
    from argparse import Namespace
    argv = str(golden['e2e.argv']).split()
    net = builtin_net('vgg19')
    weights = synthetic_weights(net, 0)
    runs = []
    for ahead in (False, True):
        state = Namespace()
        args = parse_args(state, argv, config_py=False)
        farm = TileFarm(net, [0], weights, verbose=False)
        st = StyleTransfer(farm, args, state)
        log = []
        np.random.seed(args.seed)
        callback = _BehindCallback(log) if ahead else \
            (lambda **kw: log.append((kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
        st.transfer_multiscale([Image.fromarray(golden['e2e.content_u8'])],
                               [Image.fromarray(golden['e2e.style_u8'])], callback=callback)
        runs.append((log, st.current_raw.get(), np.random.get_state()[1].copy()))
        farm.close()
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(runs[0][1], runs[1][1])
    assert np.array_equal(runs[0][2], runs[1][2])


def test_callback_that_reads_the_image_gets_the_image_of_its_step(golden):
    """--save-every under the run-ahead loop: a callback that declares (wants_image) that it will
    read iteration n's image is served before iteration n + 1 is queued -- the picture it sees is
    the one the blocking loop shows at that iteration, bit for bit."""
    from argparse import Namespace
    argv = str(golden['e2e.argv']).split()
    net = builtin_net('vgg19')
    weights = synthetic_weights(net, 0)

    class Snap:
        def __init__(self, declare):
            self.pictures, self.n = {}, 0
            if declare:
                self.wants_image = lambda n: n in (2, 4)

        def __call__(self, **kw):
            self.n += 1
            if self.n in (2, 4):
                self.pictures[self.n] = kw['transfer'].current_raw.get().copy()

    shots = []
    for declare in (False, True):          # blocking loop (unknown callback), run-ahead loop
        state = Namespace()
        args = parse_args(state, argv, config_py=False)
        farm = TileFarm(net, [0], weights, verbose=False)
        st = StyleTransfer(farm, args, state)
        snap = Snap(declare)
        np.random.seed(args.seed)
        st.transfer_multiscale([Image.fromarray(golden['e2e.content_u8'])],
                               [Image.fromarray(golden['e2e.style_u8'])], callback=snap)
        assert sorted(snap.pictures) == [2, 4]
        shots.append(snap.pictures)
        farm.close()
    for n in (2, 4):
        assert np.array_equal(shots[0][n], shots[1][n]), n


def test_device_preprocessing_equals_host_stitching():
    """prepare_features on the GPU (cut with roll offset, stx_map_place, stx_map_roll_add) must be
    bit-identical to the host-stitched version that mirrors the reference line by line."""
    net = builtin_net('vgg16_avgpool')
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    rng = np.random.RandomState(2)
    img = rng.uniform(-110, 120, (3, 150, 130)).astype(np.float32)
    layers = ['conv1_1', 'conv4_2', 'pool2']
    np.random.seed(77)
    host = farm.prepare_features(img, layers, tile_size=64, passes=4)
    np.random.seed(77)
    dev = farm.prepare_features_device(img, layers, tile_size=64, passes=4)
    for layer in layers:
        got = dev[layer].get()
        assert got.shape == host[layer].shape
        assert np.array_equal(got, host[layer]), layer
    # Gram of a device-resident map == Gram of the same map on the host
    assert np.array_equal(farm.gram_matrix(dev['conv1_1']), farm.gram_matrix(host['conv1_1']))
    farm.close()


def _run_fixture(golden, key, style_keys, kernels):
    """One multi-scale run of an end-to-end fixture on the GPU; kernels 'fp32': with the fp32-MFMA
    kernels only (gpu_helpers.fp32_kernels).  Returns (log [steps][4], final raw image, tile evaluations)."""
    import contextlib
    from argparse import Namespace
    from tests.gpu_helpers import fp32_kernels
    argv = str(golden[key + '.argv']).split()
    state = Namespace()
    args = parse_args(state, argv, config_py=False)
    net = builtin_net(args.model)
    with fp32_kernels() if kernels == 'fp32' else contextlib.nullcontext():
        farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
        st = StyleTransfer(farm, args, state)
        log = []
        np.random.seed(args.seed)
        st.transfer_multiscale([Image.fromarray(golden[key + '.content_u8'])],
                               [Image.fromarray(golden['%s.%s' % (key, k)]) for k in style_keys],
                               callback=lambda **kw: log.append(
                                   (kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
        raw = st.current_raw.get()
        evals = farm.tile_evals
        farm.close()
    import os
    if os.environ.get('STX_E2E_DUMP'):          # (for tests/golden/branch_sets.py's offline comparison)
        np.savez_compressed(os.path.join(os.environ['STX_E2E_DUMP'], '%s_%s.npz' % (key, kernels)),
                            log=np.float64(log), final_raw=raw)
    return np.float64(log), raw, evals


# The two chaotic L-BFGS fixtures run twice.  'fp32': with the fp32-MFMA kernels only (round 4's arithmetic:
# STX_CONV_H2=0, fp32 Gram and SYMM) the run is held to the COMMITTED reference run alone, with the bounds the
# tests had before the fp16-split kernels existed -- a regression anywhere else cannot hide inside a set of
# outcomes.  'default': the shipped kernels must follow ONE outcome of the reference's own code under another
# float32 implementation of its Convolution layer / Gram matrix (tests/golden/branch_sets.py), whole run and
# whole picture against that single outcome.
@pytest.mark.parametrize('kernels', ['fp32', 'default'])
def test_lbfgs_avgpool_two_styles_matches_reference_run(golden, kernels):
    """BASELINE configs 4 / 5 in miniature: -o lbfgs, vgg16_avgpool, two style images (Grams
    averaged), 2 scales, 2x2 tiles -- against the reference's own transfer_multiscale run."""
    got, raw, evals = _run_fixture(golden, 'e2e_lbfgs', ['style0_u8', 'style1_u8'], kernels)
    # L-BFGS evaluates the objective once more at the start of every scale (optimizers.py:76-77)
    assert evals == 4 * (3 + 1) + 4 * (2 + 1)
    if kernels == 'fp32':
        ref = golden['e2e_lbfgs.log']
        assert got.shape == ref.shape
        assert np.allclose(got[:, 2], ref[:, 2], rtol=2e-4), (got[:, 2], ref[:, 2])
        assert np.allclose(got[:, 1], ref[:, 1], rtol=2e-3)
        diff = np.abs(raw - golden['e2e_lbfgs.final_raw'])
        print('fp32 kernels against the committed picture: max %.4f mean %.6f' % (diff.max(), diff.mean()))
        assert diff.max() < 2.0 and diff.mean() < 0.02, (diff.max(), diff.mean())
        return
    # The shipped kernels, against the committed run ALONE.  What the reference's own code does on this fixture
    # with another float32 implementation of its Convolution layer or its Gram matrix (tests/golden/
    # branch_sets.py lbfgs: 14 + 6 real implementations, 12 runs with noise of the amplitude those measure --
    # tests/golden/lbfgs_runs.json, one picture per distinct outcome in lbfgs_branches.npz): the losses stay
    # within 8.2e-4 (30 of 33 runs within 1e-5), 30 runs reproduce the committed picture to 0.001, the other
    # three move ONE patch of it -- a ReLU near-tie of a 40-pixel tile's deep layers decided the other way,
    # amplified by five L-BFGS steps -- by 0.59 (3 values beyond 0.5), 1.82 (71) and 7.03 (779; mean 0.12).
    # So: the losses to the fixture's 2e-4, the mean to the bound it always had, and the displaced patch
    # bounded in height AND in area by what the reference does to itself -- instead of round 5's per-pixel
    # minimum over a set of outcomes (a patchwork of outcomes passed that).
    from tests.helpers import lbfgs_reference_outcomes
    outcomes = lbfgs_reference_outcomes(golden)
    ref = golden['e2e_lbfgs.log']
    assert got.shape == ref.shape
    assert np.allclose(got[:, 2], ref[:, 2], rtol=2e-4), (got[:, 2], ref[:, 2])
    assert np.allclose(got[:, 1], ref[:, 1], rtol=2e-3)
    diff = np.abs(raw - golden['e2e_lbfgs.final_raw'])
    own = [np.abs(o['final_raw'] - golden['e2e_lbfgs.final_raw']) for o in outcomes[1:]]
    print('against the committed picture: max %.4f mean %.6f, %d values beyond 0.5 (the reference against itself: %s)'
          % (diff.max(), diff.mean(), (diff > 0.5).sum(),
             ', '.join('%.2f / %d' % (d.max(), (d > 0.5).sum()) for d in own)))
    assert diff.mean() < 0.02, (diff.max(), diff.mean())
    assert diff.max() <= max(d.max() for d in own) + 0.5, diff.max()
    assert (diff > 0.5).sum() <= 0.005 * diff.size, int((diff > 0.5).sum())       # one patch: 0.5 % of the picture


@pytest.mark.parametrize('kernels', ['fp32', 'default'])
def test_config4_miniature_matches_reference_run(golden, kernels):
    """BASELINE config 4's own combination -- VGG-19 (MAX pooling) x L-BFGS x a 3 x 3 tiling whose
    last row and column are larger (style_transfer.py:619-632) -- against the reference's run
    (tests/golden/make_golden.py section 4c): 65 x 71 with 2 x 2 tiles, then 92 x 100 with tiles
    of 30/30/32 x 33/33/34."""
    got, raw, evals = _run_fixture(golden, 'e2e_cfg4', ['style_u8'], kernels)
    # 4 tiles x (3 + 1) evaluations at the first scale, 9 tiles x (2 + 1) at the second
    assert evals == 4 * (3 + 1) + 9 * (2 + 1)
    from tests.helpers import cfg4_reference_branches, matching_branch
    branches = cfg4_reference_branches(golden)
    print([np.array2string(got[:, 2] / b['log'][:, 2] - 1, precision=2) for b in branches])
    assert got.shape == branches[0]['log'].shape
    # fp32 kernels: the committed trajectory and no other.  Default kernels: the reference's trajectory on this
    # fixture BRANCHES under another float32 implementation of its convolutions (tests/golden/branch_sets.py
    # cfg4: tiles of 30 x 33 pixels, one near-tie moves the objective by 1e-3) -- ONE of its branches, to 2e-4.
    br = matching_branch(branches[:1] if kernels == 'fp32' else branches, got[:, 2])
    assert br is not None, (got[:, 2], [b['log'][:, 2] for b in branches])
    ref = br['log']
    assert np.allclose(got[:, 1], ref[:, 1], rtol=2e-3)
    assert np.allclose(got[:, 3], ref[:, 3], rtol=2e-3)
    # (the last line search flips on its own: the unperturbed reference repeated differs from itself by
    # 0.06 .. 0.9 in the final picture)
    diff = np.abs(raw - br['final_raw'])
    print('final image: max %.4f mean %.6f (branch taken by %d of the reference\'s runs)'
          % (diff.max(), diff.mean(), br['runs']))
    assert diff.max() < 2.0 and diff.mean() < 0.02, (diff.max(), diff.mean())


@pytest.mark.parametrize('kernels', ['fp32', 'default'])
def test_lbfgs_multi_tile_run_at_a_stable_size_matches_reference_run(golden, kernels):
    """The L-BFGS pin at a size where ONE reference trajectory exists (make_golden.py
    section 4j): VGG-19 with AVE pooling x L-BFGS x a ragged 2 x 2 tiling of 97 .. 140-pixel tiles, 3 + 2
    iterations.  tests/golden/branch_sets.py stable: under every other float32 implementation of the
    reference's Convolution layer tried (14) and calibrated noise (8 runs) its losses stay within 4e-5 of
    the committed run at every step -- so the shipped kernels are held to that one trajectory at the plain
    2e-4.  (With MAX pooling the reference leaves the band at ANY size: the 280-pixel VGG-19 run moves by
    2e-4 at step 2 and 2.4e-3 at step 5 between its own SGEMM and torch's conv2d.)  The final picture still
    carries the ReLU near-ties of five steps as displaced patches (the reference against itself: max 9 .. 20,
    mean 0.01 .. 0.07): bounded in the mean."""
    got, raw, evals = _run_fixture(golden, 'e2e_stable', ['style_u8'], kernels)       # (both kernel sets: one trajectory)
    assert evals == 4 * (3 + 1) + 4 * (2 + 1)
    ref = golden['e2e_stable.log']
    assert got.shape == ref.shape
    print('stable fixture: loss against the committed run', np.array2string(got[:, 2] / ref[:, 2] - 1, precision=2))
    assert np.allclose(got[:, 2], ref[:, 2], rtol=2e-4), (got[:, 2], ref[:, 2])
    assert np.allclose(got[:, 1], ref[:, 1], rtol=2e-3)
    assert np.allclose(got[:, 3], ref[:, 3], rtol=2e-3)
    diff = np.abs(raw - golden['e2e_stable.final_raw'])
    print('  final picture: max %.3f mean %.5f' % (diff.max(), diff.mean()))
    assert diff.mean() < 0.15, (diff.max(), diff.mean())


def test_farm_staged_leg_equals_direct_leg():
    """The multi-GPU leg of TileFarm.eval_sc_grad -- master-side staging buffers, the worker's
    pull of its tile, the master's pull of the gradient (xGMI peer copies on an 8-GPU node) --
    forced on ONE GPU: engines 1..3 are treated as remote.  Loss and stitched gradient must be
    bit-identical to the direct leg (style_transfer.py:284-288,634-643: round-robin tiles,
    disjoint stitch)."""
    from style_transfer_amd import image_ops
    net = builtin_net('vgg19')
    weights = synthetic_weights(net, 0)
    rng = np.random.RandomState(21)
    img = rng.uniform(-110, 120, (3, 150, 170)).astype(np.float32)
    style = rng.uniform(-110, 120, (3, 64, 72)).astype(np.float32)
    cl, cw = ['conv4_2'], {'conv4_2': 0.05}
    sl = ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
    sw = {l: 0.2 for l in sl}
    results = []
    for staged in (False, True):
        farm = TileFarm(net, [0], weights, verbose=False, force_staging=staged, streams_per_device=4)
        np.random.seed(3)
        contents = [farm.prepare_features_device(img, cl, 64, passes=2)]
        feats = farm.prepare_features_device(style, sl, 64, passes=1)
        styles = [{l: farm.gram_matrix(f) for l, f in feats.items()}]
        farm.set_contents_and_styles(contents, styles)
        eng = farm.master
        d_img, d_grad = eng.to_device(img), eng.empty(img.shape).zero()
        loss = farm.eval_sc_grad(d_img, d_grad, (24, -16), cl, sl, {}, cw, sw, 64)   # 3 x 3 tiles
        assert len(farm.engines) == 4 and farm.tile_evals == 9
        assert bool(farm._staging) == staged
        results.append((loss, d_grad.get()))
        farm.close()
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])


def test_aux_image_term_matches_reference_run(golden):
    """--aux-image (style_transfer.py:729-733): the reference compares the ROLLED image with the
    un-rolled auxiliary image; stx_image_regularizers reproduces that through aux_roll_xy."""
    from argparse import Namespace
    argv = str(golden['e2e_aux.argv']).split()
    state = Namespace()
    args = parse_args(state, argv, config_py=False)
    net = builtin_net(args.model)
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    st = StyleTransfer(farm, args, state)
    log = []
    np.random.seed(args.seed)
    st.transfer_multiscale([Image.fromarray(golden['e2e_aux.content_u8'])],
                           [Image.fromarray(golden['e2e_aux.style_u8'])],
                           aux_image=Image.fromarray(golden['e2e_aux.aux_u8']),
                           callback=lambda **kw: log.append(
                               (kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
    ref, got = golden['e2e_aux.log'], np.float64(log)
    assert got.shape == ref.shape
    # (with the auxiliary image left un-rolled the losses of steps 2 and 3 are off by 4e-3 and
    # 5e-3 and the final image by 55: measured, tools/aux_check.py)
    assert np.allclose(got[:, 2], ref[:, 2], rtol=3e-4), (got[:, 2], ref[:, 2])
    # Adam's first steps move every pixel by +-step_size whatever the size of its gradient, so a
    # pixel whose summed gradient is within float noise of zero may go the other way: isolated
    # pixels differ by several units, everything else agrees closely
    diff = np.abs(st.current_raw.get() - golden['e2e_aux.final_raw'])
    print('aux run: max %.3f mean %.5f p99.9 %.4f' % (diff.max(), diff.mean(),
                                                      np.percentile(diff, 99.9)))
    assert diff.mean() < 0.05 and np.percentile(diff, 99) < 0.5, (diff.max(), diff.mean())
    farm.close()


def test_jitter_and_deep_dream_run_matches_reference(golden):
    """--jitter (content maps recomputed every iteration from the shifted picture, any pixel shift:
    style_transfer.py:757-763,780-794) with --dd-weight / --dd-layers, against the reference's
    own transfer_multiscale run."""
    from argparse import Namespace
    argv = str(golden['e2e_jitter.argv']).split()
    state = Namespace()
    args = parse_args(state, argv, config_py=False)
    net = builtin_net(args.model)
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    st = StyleTransfer(farm, args, state)
    log = []
    np.random.seed(args.seed)
    st.transfer_multiscale([Image.fromarray(golden['e2e_jitter.content_u8'])],
                           [Image.fromarray(golden['e2e_jitter.style_u8'])],
                           callback=lambda **kw: log.append(
                               (kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
    ref, got = golden['e2e_jitter.log'], np.float64(log)
    print(got[:, 2] / ref[:, 2] - 1)
    assert got.shape == ref.shape
    assert np.allclose(got[:, 2], ref[:, 2], rtol=3e-4), (got[:, 2], ref[:, 2])
    diff = np.abs(st.current_raw.get() - golden['e2e_jitter.final_raw'])
    print('jitter run: max %.3f mean %.5f' % (diff.max(), diff.mean()))
    assert diff.mean() < 0.05 and np.percentile(diff, 99) < 0.5, (diff.max(), diff.mean())
    farm.close()


def test_swt_weight_runs_and_adds_its_term(golden):
    """--swt-weight with the default wavelet / level count (style_transfer.py:716-720): the run
    goes through, and on the first evaluation -- same seed, so same start image and same shift --
    the loss exceeds the run without it by the SWT term of that image (the term itself is held to
    the oracle in tests/test_gpu_image_ops.py; PyWavelets is absent, so no reference run exists)."""
    from argparse import Namespace
    base = str(golden['e2e_aux.argv']).split()
    losses = []
    for extra in ([], ['--swt-weight', '3']):
        state = Namespace()
        args = parse_args(state, base + extra + ['-i', '2'], config_py=False)
        net = builtin_net(args.model)
        farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
        st = StyleTransfer(farm, args, state)
        log = []
        np.random.seed(args.seed)
        st.transfer_multiscale([Image.fromarray(golden['e2e_aux.content_u8'])],
                               [Image.fromarray(golden['e2e_aux.style_u8'])],
                               callback=lambda **kw: log.append(kw['loss']))
        losses.append(log)
        farm.close()
    assert all(np.isfinite(l) for run in losses for l in run)
    assert losses[1][0] > losses[0][0]            # first evaluation: identical image, one more term


def test_style_multiscale_run_matches_reference(golden, capsys):
    """--style-multiscale MIN MAX (style_transfer.py:493-531): the style Gram is the mean over a
    sqrt(2) ladder of resamplings of the style picture -- here three (48, 68, 96) -- against the
    reference's own transfer_multiscale run, the ladder it announced included."""
    from argparse import Namespace
    argv = str(golden['e2e_sm.argv']).split()
    state = Namespace()
    args = parse_args(state, argv, config_py=False)
    net = builtin_net(args.model)
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    st = StyleTransfer(farm, args, state)
    log = []
    np.random.seed(args.seed)
    st.transfer_multiscale([Image.fromarray(golden['e2e_sm.content_u8'])],
                           [Image.fromarray(golden['e2e_sm.style_u8'])],
                           callback=lambda **kw: log.append(
                               (kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
    printed = [l for l in capsys.readouterr().out.splitlines() if l.startswith('Processing style')]
    assert printed == str(golden['e2e_sm.style_lines']).splitlines()
    ref, got = golden['e2e_sm.log'], np.float64(log)
    assert got.shape == ref.shape
    assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-4), (got[:, 2], ref[:, 2])
    assert np.allclose(got[:, 3], ref[:, 3], rtol=1e-4)
    diff = np.abs(st.current_raw.get() - golden['e2e_sm.final_raw'])
    print('style-multiscale run: max %.3f mean %.5f' % (diff.max(), diff.mean()))
    assert diff.mean() < 0.05 and np.percentile(diff, 99) < 0.5, (diff.max(), diff.mean())
    farm.close()


def test_many_options_run_matches_reference(golden, capsys):
    """One reference run that leaves the defaults in many places at once: --init-image (Adam's
    biased first-moment start, style_transfer.py:883-897), --style-scale, --div, --step-decay,
    --avg-window, --tv-power 1.5, --p-power 4, --mean, weighted non-default content and style
    layers; two scales, tiled."""
    from argparse import Namespace
    argv = str(golden['e2e_opts.argv']).split()
    state = Namespace()
    args = parse_args(state, argv, config_py=False)
    net = builtin_net(args.model)
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    st = StyleTransfer(farm, args, state)
    log = []
    np.random.seed(args.seed)
    st.transfer_multiscale([Image.fromarray(golden['e2e_opts.content_u8'])],
                           [Image.fromarray(golden['e2e_opts.style_u8'])],
                           initial_image=Image.fromarray(golden['e2e_opts.init_u8']),
                           callback=lambda **kw: log.append(
                               (kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
    want = [l for l in str(golden['e2e_opts.lines']).splitlines() if l.startswith('Scale ')]
    got_lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith('Scale ')]
    assert got_lines == want
    ref, got = golden['e2e_opts.log'], np.float64(log)
    print(got, ref)
    assert got.shape == ref.shape
    assert np.array_equal(got[:, 0], ref[:, 0])
    assert np.allclose(got[:, 2], ref[:, 2], rtol=2e-4), (got[:, 2], ref[:, 2])
    assert np.allclose(got[:, 3], ref[:, 3], rtol=2e-4), (got[:, 3], ref[:, 3])
    assert np.allclose(got[:, 1], ref[:, 1], rtol=2e-3), (got[:, 1], ref[:, 1])
    diff = np.abs(st.current_raw.get() - golden['e2e_opts.final_raw'])
    print('many-options run: max %.3f mean %.5f' % (diff.max(), diff.mean()))
    assert diff.mean() < 0.05 and np.percentile(diff, 99) < 0.5, (diff.max(), diff.mean())
    farm.close()


@pytest.mark.parametrize('tag', ['abs', 'max'])
def test_style_size_rules_match_reference(golden, tag):
    """The remaining branches of the style-size rule (style_transfer.py:862-872): an absolute
    --style-scale (>= 32: fitted into that many pixels, scaled up if need be) and --max-style-size
    with --style-scale-up, against the reference's own runs."""
    from argparse import Namespace
    argv = str(golden['e2e_ss.%s.argv' % tag]).split()
    state = Namespace()
    args = parse_args(state, argv, config_py=False)
    net = builtin_net(args.model)
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    st = StyleTransfer(farm, args, state)
    log = []
    np.random.seed(args.seed)
    st.transfer_multiscale([Image.fromarray(golden['e2e_ss.content_u8'])],
                           [Image.fromarray(golden['e2e_ss.style_u8'])],
                           callback=lambda **kw: log.append(
                               (kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
    ref, got = golden['e2e_ss.%s.log' % tag], np.float64(log)
    assert got.shape == ref.shape
    assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-4), (got[:, 2], ref[:, 2])
    diff = np.abs(st.current_raw.get() - golden['e2e_ss.%s.final_raw' % tag])
    assert diff.mean() < 0.05 and np.percentile(diff, 99) < 0.5, (diff.max(), diff.mean())
    farm.close()
