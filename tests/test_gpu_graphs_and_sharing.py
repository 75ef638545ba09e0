"""Round-3 host machinery on the GPU: recorded launch graphs (stx_sc_grad_tile replayed as one
hipGraph), per-GPU sharing of weights / packed banks / targets between the engines of a farm, and
the event-ordered scatter / gather of TileFarm.eval_sc_grad.  None of it may change a bit of the
results: every check here is `==` against the kernel-by-kernel path."""

import numpy as np
import pytest

from tests.gpu_helpers import builtin_net, require_gpu, synthetic_weights

pytestmark = pytest.mark.gpu

CL, SL = ['conv4_2'], ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
CW, SW = {'conv4_2': 0.05}, {l: 0.2 for l in SL}


def _targets(eng, full_hw, rng):
    h, w = full_hw
    contents = [{l: np.abs(rng.standard_normal(eng.feature_shape(l, h, w))).astype(np.float32)
                 for l in CL}]
    styles = [{l: np.tril(rng.standard_normal((eng.layer_info(l)[1],) * 2)).astype(np.float32)
               for l in SL}]
    return contents, styles


@pytest.mark.parametrize('model', ['vgg19', 'vgg16_avgpool'])
def test_replayed_graph_equals_eager_calls(model, monkeypatch):
    """The same sequence of tile evaluations -- different tiles, tile origins and seam-suppression
    shifts from call to call, two tile shapes interleaved -- with and without recorded launch
    graphs: identical losses and gradients, and the recordings were actually replayed.  The
    shift and the origin are the per-call state a recording cannot bake in (ContentWindow::dyn);
    style_transfer.py:571-573,647-655."""
    from style_transfer_amd import lib
    from style_transfer_amd.engine import TileEngine
    require_gpu()
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(5)
    full = (160, 192)
    shapes = [(64, 96), (96, 64)]
    calls = []
    for i in range(14):
        th, tw = shapes[i % 2]
        start = (8 * rng.randint(0, (full[0] - th) // 8 + 1), 8 * rng.randint(0, (full[1] - tw) // 8 + 1))
        roll = (8 * rng.randint(-30, 30), 8 * rng.randint(-30, 30))
        calls.append((th, tw, start, roll, rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)))
    results = {}
    monkeypatch.setenv('STX_GRAPH_MIN_EAGER', '2')      # (default: 300 evaluations before a recording)
    monkeypatch.setenv('STX_SIDE_STREAM', '0')          # (the default; a second stream is never recorded)
    for mode in ('0', '1'):
        monkeypatch.setenv('STX_GRAPH', mode)
        eng = TileEngine(net, 0, weights)
        eng.set_contents_and_styles(*_targets(eng, full, np.random.RandomState(9)))
        bufs = {s: (eng.empty((3,) + s), eng.empty((3,) + s)) for s in shapes}
        out = []
        for th, tw, start, roll, tile in calls:
            d_tile, d_grad = bufs[(th, tw)]
            d_tile.set(tile)
            p = eng.sc_grad_tile_async(d_tile, start, roll, CL, SL, {'conv3_1': 1.5}, CW, SW,
                                       grad_out=d_grad)
            eng.sync()
            out.append((p.loss, d_grad.get().copy()))
        counters = (eng.query(lib.Q_GRAPH_CAPTURES), eng.query(lib.Q_GRAPH_REPLAYS),
                    eng.query(lib.Q_EAGER_TILES))
        results[mode] = (out, counters)
        eng.close()
    assert results['0'][1][:2] == (0, 0) and results['0'][1][2] == len(calls)
    captures, replays, eager = results['1'][1]
    # per shape: two eager evaluations (one more if a buffer grew in between), then the recording
    assert captures == 2 and replays + eager == len(calls) and replays >= len(calls) - 6, results['1'][1]
    for (la, ga), (lb, gb) in zip(results['0'][0], results['1'][0]):
        assert la == lb
        assert np.array_equal(ga, gb)


def test_graph_instances_between_syncs_and_rerecording_after_new_targets(monkeypatch):
    """Several evaluations of one key between two syncs each own their loss scalars (separate
    recordings), and new targets (a new scale: every device pointer may move) retire the old
    recordings instead of replaying them."""
    from style_transfer_amd import lib
    from style_transfer_amd.engine import TileEngine
    require_gpu()
    monkeypatch.setenv('STX_GRAPH_MIN_EAGER', '2')
    monkeypatch.setenv('STX_SIDE_STREAM', '0')
    net = builtin_net('vgg19')
    eng = TileEngine(net, 0, synthetic_weights(net.as_dicts(), 0))
    rng = np.random.RandomState(2)
    th = tw = 64
    tiles = [eng.to_device(rng.uniform(-110, 120, (3, th, tw)).astype(np.float32)) for _ in range(3)]
    grads = [eng.empty((3, th, tw)) for _ in range(3)]

    def run_round():
        pend = [eng.sc_grad_tile_async(t, (0, 0), (0, 0), CL, SL, {}, CW, SW, grad_out=g)
                for t, g in zip(tiles, grads)]
        eng.sync()
        return [p.loss for p in pend], [g.get().copy() for g in grads]

    for scale in range(2):
        eng.set_contents_and_styles(*_targets(eng, (th, tw), np.random.RandomState(scale)))
        rounds = [run_round() for _ in range(5)]
        for losses, grad_arrays in rounds[1:]:
            assert losses == rounds[0][0]
            assert all(np.array_equal(a, b) for a, b in zip(grad_arrays, rounds[0][1]))
        assert len(set(rounds[0][0])) == 3          # three different tiles, three losses
    # per scale: 3 instances x 2 eager evaluations, 3 recordings, 3 x 3 replays
    assert eng.query(lib.Q_GRAPH_CAPTURES) == 6
    assert eng.query(lib.Q_EAGER_TILES) + eng.query(lib.Q_GRAPH_REPLAYS) == 30
    assert eng.query(lib.Q_GRAPH_REPLAYS) >= 15
    eng.close()


def test_engines_of_one_gpu_share_weights_banks_and_targets():
    """A farm's engines on one GPU hold ONE copy of the weights, the packed filter banks and the
    targets, and the targets are uploaded once per GPU and scale (the reference sends them to
    every worker process: style_transfer.py:309-332)."""
    from style_transfer_amd import lib
    from style_transfer_amd.engine import TileEngine
    from style_transfer_amd.farm import TileFarm
    require_gpu()
    net = builtin_net('vgg19')
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(21)
    img = rng.uniform(-110, 120, (3, 128, 128)).astype(np.float32)

    def one_step(farm):
        eng = farm.master
        np.random.seed(3)
        contents = [farm.prepare_features_device(img, CL, 64, passes=2)]
        feats = farm.prepare_features_device(img[:, :64, :72], SL, 64, passes=1)
        farm.set_contents_and_styles(contents, [{l: farm.gram_matrix(f) for l, f in feats.items()}])
        d_img, d_grad = eng.to_device(img), eng.empty(img.shape).zero()
        loss = farm.eval_sc_grad(d_img, d_grad, (16, -8), CL, SL, {}, CW, SW, 64)     # 2 x 2 tiles
        return loss, d_grad.get()

    solo = TileFarm(net, [0], weights, verbose=False, streams_per_device=1)
    ref = one_step(solo)
    solo_bytes = solo.master.query(lib.Q_WEIGHT_BYTES)
    assert solo.master.query(lib.Q_SHARED_ENGINES) == 1
    solo.close()

    farm = TileFarm(net, [0], weights, verbose=False, streams_per_device=4)
    got = one_step(farm)
    assert len(farm.engines) == 4 and len(farm.primaries()) == 1
    for eng in farm.engines:
        assert eng.query(lib.Q_SHARED_ENGINES) == 4
        assert eng.query(lib.Q_TARGET_UPLOADS) == 1          # one upload for four engines
        assert eng.query(lib.Q_WEIGHT_BYTES) == solo_bytes   # one bank, not four
    one_step(farm)
    assert farm.master.query(lib.Q_TARGET_UPLOADS) == 2
    assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
    # a stand-alone engine is its own group
    other = TileEngine(net, 0, weights)
    assert other.query(lib.Q_SHARED_ENGINES) == 1 and other.query(lib.Q_TARGET_UPLOADS) == 0
    other.close()
    farm.close()


def test_lazy_loss_and_stream_ordered_gradient(monkeypatch):
    """eval_sc_grad(lazy=True) returns without a host wait; the gradient is complete in stream
    order on the master and float(loss) equals the synchronous call."""
    from style_transfer_amd.farm import LazyLoss, TileFarm
    require_gpu()
    monkeypatch.setenv('STX_GRAPH_MIN_EAGER', '2')
    monkeypatch.setenv('STX_SIDE_STREAM', '0')
    net = builtin_net('vgg19')
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(4)
    img = rng.uniform(-110, 120, (3, 96, 160)).astype(np.float32)
    farm = TileFarm(net, [0], weights, verbose=False, force_staging=True)
    eng = farm.master
    contents = [farm.prepare_features_device(img, CL, 64, passes=1)]
    feats = farm.prepare_features_device(img[:, :64, :64], SL, 64, passes=1)
    farm.set_contents_and_styles(contents, [{l: farm.gram_matrix(f) for l, f in feats.items()}])
    d_img, g_sync, g_lazy = eng.to_device(img), eng.empty(img.shape).zero(), eng.empty(img.shape).zero()
    want = farm.eval_sc_grad(d_img, g_sync, (8, 24), CL, SL, {}, CW, SW, 64)
    for _ in range(4):                      # eager, eager, recorded, replayed
        lazy = farm.eval_sc_grad(d_img, g_lazy, (8, 24), CL, SL, {}, CW, SW, 64, lazy=True)
        assert isinstance(lazy, LazyLoss)
        copy = eng.empty(img.shape).copy_from(g_lazy)       # ordered behind the stitch on the master
        assert float(lazy) == want
        assert np.array_equal(copy.get(), g_sync.get())
        copy.free()
    farm.close()


def test_zero_copy_tiles_equal_copied_tiles():
    """One tile per engine: the master cuts straight into the engines' input blobs and stitches out
    of their gradient blobs (stx_tile_buffers).  Same bits as with separate tile buffers and the
    two device-to-device copies per tile."""
    from style_transfer_amd.farm import TileFarm
    require_gpu()
    net = builtin_net('vgg19')
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(8)
    img = rng.uniform(-110, 120, (3, 128, 144)).astype(np.float32)
    results = []
    for zero_copy in (True, False):
        farm = TileFarm(net, [0], weights, verbose=False)
        farm.zero_copy = zero_copy
        eng = farm.master
        np.random.seed(1)
        contents = [farm.prepare_features_device(img, CL, 96, passes=2)]
        feats = farm.prepare_features_device(img[:, :64, :64], SL, 96, passes=1)
        farm.set_contents_and_styles(contents, [{l: farm.gram_matrix(f) for l, f in feats.items()}])
        d_img, d_grad = eng.to_device(img), eng.empty(img.shape).zero()
        losses = [farm.eval_sc_grad(d_img, d_grad, (8 * k, -16), CL, SL, {}, CW, SW, 96) for k in range(3)]
        assert farm.tile_evals == 12 and bool(farm._tiles) != zero_copy
        results.append((losses, d_grad.get()))
        farm.close()
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])


@pytest.mark.parametrize('model', ['vgg19', 'vgg16_avgpool'])
def test_loss_terms_on_a_second_stream_change_nothing(model, monkeypatch):
    """Tiles of up to 512 x 512 run Gram / SYMM / content sums on a second HIP stream beside the
    rest of the forward and the backward pass (they cannot fill the chip on their own); the
    backward walk waits for each tap's event.  Same kernels, same order of additions: identical
    to the single-stream schedule, repeatedly (a missing dependency would show as a race)."""
    from style_transfer_amd.engine import TileEngine
    require_gpu()
    net = builtin_net(model)
    weights = synthetic_weights(net.as_dicts(), 0)
    rng = np.random.RandomState(12)
    th, tw = 136, 200
    tiles = [rng.uniform(-110, 120, (3, th, tw)).astype(np.float32) for _ in range(3)]
    results = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('STX_SIDE_STREAM', mode)
        eng = TileEngine(net, 0, weights)
        eng.set_contents_and_styles(*_targets(eng, (th + 8, tw), np.random.RandomState(3)))
        out = []
        for rep in range(4):
            for tile in tiles:
                loss, grad = eng.sc_grad_tile(tile, (8, 0), (16, -8), CL, SL, {'conv3_1': 0.5}, CW, SW)
                out.append((loss, grad.copy()))
        results[mode] = out
        eng.close()
    for (la, ga), (lb, gb) in zip(results['0'], results['1']):
        assert la == lb
        assert np.array_equal(ga, gb)
