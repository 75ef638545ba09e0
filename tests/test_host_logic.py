"""Host-side logic (no GPU): options, tile geometry, pyramid, weight parsing, model files."""

import ast
import os

import numpy as np
import pytest
from PIL import Image

from oracle import tile_path
from style_transfer_amd import config_system, farm, netspec, transfer, weights


def test_parse_args_defaults_match_reference(golden):
    """Every option the reference's parse_args produces exists here with the same default."""
    ref = dict(ast.literal_eval(str(golden['args.defaults_repr'])))
    args = config_system.parse_args(None, ['-ci', 'c.png', '-si', 's.png'], config_py=False)
    mine = {k: str(getattr(args, k)) for k in args}
    assert set(ref) == set(mine)
    for key, value in ref.items():
        assert mine[key] == value, key


def test_option_precedence_and_lazy_values(tmp_path):
    cfg1 = tmp_path / 'config.py'
    cfg1.write_text('size = 640\ntile_size = 320\ntv_weight = lambda s: 5 / (s.step + 1)\n')
    cfg2 = tmp_path / 'extra.py'
    cfg2.write_text('tile_size = 200\ndevices = detect_devices()\n')
    from argparse import Namespace
    state = Namespace(step=4)
    args = config_system.parse_args(state, ['-ci', 'c', '-si', 's', '--size', '300', '--config',
                                            str(cfg2)], config_py=cfg1)
    assert args.size == 300          # command line beats config.py
    assert args.tile_size == 200     # --config file beats everything
    assert args.tv_weight == 1.0     # callable of the state, evaluated on read
    state.step = 0
    assert args.tv_weight == 5.0
    assert args.devices in ([-1], list(range(len(args.devices))))
    assert config_system.ffloat('1/4') == 0.25


@pytest.mark.parametrize('hw,tile', [((96, 112), 64), ((75, 93), 48), ((2048, 2048), 1024),
                                     ((2896, 2896), 1024), ((1448, 1448), 1024), ((130, 150), 512),
                                     ((1024, 768), 512)])
def test_tile_grid_matches_oracle(hw, tile):
    rects = farm.tile_grid(hw, tile)
    assert rects == tile_path.tile_grid(hw, tile)
    cover = np.zeros(hw, int)
    for y0, y1, x0, x1 in rects:
        cover[y0:y1, x0:x1] += 1
        assert y1 - y0 <= tile + tile // 2 and x1 - x0 <= tile + tile // 2
    assert np.all(cover == 1)


def test_pyramid_and_resize():
    assert transfer.pyramid_sizes(1024, 182) == [1024, 724, 512, 362, 256]
    assert transfer.pyramid_sizes(2048, 182) == [2048, 1448, 1024, 724, 512, 362, 256]
    assert transfer.pyramid_sizes(4096, 182)[:3] == [4096, 2896, 2048]
    img = Image.new('RGB', (400, 300))
    assert transfer.resize_to_fit(img, 200).size == (200, 150)
    assert transfer.resize_to_fit(img, 800).size == (400, 300)            # never up...
    assert transfer.resize_to_fit(img, 800, scale_up=True).size == (800, 600)   # ...unless asked
    assert transfer.resize_to_fit(img, 250, div=32).size == (224, 160)


def test_parse_weights_normalises_to_master():
    names, w = transfer.parse_weights(['conv1_1', 'conv2_1:3', 'conv3_1:1/2'], 2.0)
    assert names == ['conv1_1', 'conv2_1', 'conv3_1']
    assert sum(abs(v) for v in w.values()) == pytest.approx(2.0)
    assert w['conv2_1'] == pytest.approx(3 * w['conv1_1'])


@pytest.mark.parametrize('legacy', [False, True])
def test_caffemodel_round_trip(tmp_path, legacy):
    net = netspec.builtin_net('vgg16')
    params = weights.synthetic_weights(net, 4)
    small = {k: params[k] for k in list(params)[:4]}
    path = str(tmp_path / 'w.caffemodel')
    weights.write_caffemodel(path, small, legacy=legacy)
    short = netspec.NetSpec('short', [l for l in net.layers if l.name in
                                      ('input', 'conv1_1', 'relu1_1', 'conv1_2', 'relu1_2', 'pool1',
                                       'conv2_1', 'relu2_1', 'conv2_2', 'relu2_2')])
    loaded = weights.load_weights(path, short)
    for name, (w, b) in small.items():
        assert np.array_equal(loaded[name][0], w) and np.array_equal(loaded[name][1], b)
    with pytest.raises(FileNotFoundError):
        weights.load_weights(str(tmp_path / 'missing.caffemodel'), short)
    assert np.array_equal(weights.load_weights('synthetic:4', net)['conv3_1'][0], params['conv3_1'][0])


def test_synthetic_weights_agree_with_oracle():
    from oracle.caffe_net import synthetic_weights as oracle_weights
    net = netspec.builtin_net('vgg19')
    a, b = weights.synthetic_weights(net, 0), oracle_weights(net.as_dicts(), 0)
    assert all(np.array_equal(a[k][0], b[k][0]) and np.array_equal(a[k][1], b[k][1]) for k in a)


def test_unsupported_options_fail_loudly():
    from argparse import Namespace
    args = config_system.parse_args(None, ['-ci', 'c', '-si', 's', '--swt-weight', '1', '--swt-wavelet', 'db2'],
                                    config_py=False)

    class FakeFarm:
        master = None

        def layers(self):
            return []
    with pytest.raises(NotImplementedError):
        transfer.StyleTransfer(FakeFarm(), args, Namespace())


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under style_transfer_amd/ may import it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        'style_transfer_amd')
    for dirpath, _, files in os.walk(root):
        for name in files:
            if name.endswith('.py'):
                text = open(os.path.join(dirpath, name)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, name


@pytest.mark.parametrize('hw,method', [((52, 75), 'lanczos'), ((26, 40), 'lanczos'),
                                       ((52, 75), 'bilinear'), ((19, 53), 'bilinear')])
def test_resample_coefficients_reproduce_pillow(hw, method):
    from style_transfer_amd import resample
    rng = np.random.RandomState(0)
    a = rng.uniform(-100, 100, (3, 37, 53)).astype(np.float32)
    pil_method = Image.LANCZOS if method == 'lanczos' else Image.BILINEAR
    ref = np.stack([np.asarray(Image.fromarray(a[c]).resize((hw[1], hw[0]), pil_method))
                    for c in range(3)])
    assert np.array_equal(resample.resample_host(a, hw, method), ref)


def test_schedule_is_planned_up_front():
    """plan_scales: pyramid sizes, per-level picture and style sizes, iterations and tile grids of
    BASELINE configs 3 and 4 (style_transfer.py:832-909,619-632)."""
    args = config_system.parse_args(None, ['-ci', 'c', '-si', 's', '--size', '2048', '--tile-size',
                                           '1024'], config_py=False)
    plans = transfer.plan_scales(args, (2048, 2048), [(1500, 1000)])
    assert [p.size for p in plans] == [256, 362, 512, 724, 1024, 1448, 2048]
    assert [p.iterations for p in plans] == [200] + [100] * 6
    assert [p.tiles for p in plans] == [(1, 1)] * 5 + [(2, 2)] * 2
    assert sum(p.iterations * p.tiles[0] * p.tiles[1] for p in plans) == 1400     # tile-iterations
    assert plans[0].style_fit == [(256, 171)] and plans[-1].style_fit == [None]   # never scaled up
    args = config_system.parse_args(None, ['-ci', 'c', '-si', 's', '--size', '4096', '--tile-size',
                                           '1024', '-o', 'lbfgs'], config_py=False)
    plans = transfer.plan_scales(args, (4096, 4096), [(4096, 4096)])
    assert [p.tiles for p in plans][-2:] == [(3, 3), (4, 4)] and plans[-2].content_wh == (2896, 2896)
    assert transfer.fit_size((400, 300), 800) is None
    assert transfer.fit_size((400, 300), 250, div=32) == (224, 160)
    args = config_system.parse_args(None, ['-ci', 'c', '-si', 's', '--style-multiscale', '100',
                                           '400'], config_py=False)
    assert transfer.style_pyramid(args) == [400, 283, 200, 141, 100]


def test_bench_per_kernel_accounting():
    """bench.py's roofline.per_kernel: launch groups are sorted into kernels by layer shape and tap, an
    fp16-split layer issues 2 x its direct FLOP count, fractions are taken against the fp16 MFMA peak."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location('stx_bench', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    sys.modules['stx_bench'] = bench
    spec.loader.exec_module(bench)
    net = netspec.builtin_net('vgg19')
    chans, scale, c_in, s = {}, {}, 3, 1
    for lay in net.as_dicts():
        if lay['type'] == 'Convolution':
            chans[lay['top']], scale[lay['top']], c_in = (c_in, lay['num_output']), s, lay['num_output']
        elif lay['type'] == 'Pooling':
            s *= 2
    convs = [l for l in chans if int(l[4]) < 5 or l == 'conv5_1']

    class Array:
        def free(self):
            pass

    class Eng:
        def __init__(self):
            self.reads = 0

        def to_device(self, host):
            return Array()

        def empty(self, shape):
            return Array()

        def sc_grad_tile_async(self, *args, **kwargs):
            pass

        def sync(self):
            pass

        def profile(self, on):
            pass

        def layer_info(self, layer):
            return scale[layer], chans[layer][1]

        def profile_read(self):
            rows = []
            for kind in ('fwd', 'bwd'):
                for l in (convs if kind == 'fwd' else convs[::-1]):
                    cin, cout = chans[l]
                    rows.append(('%s %s' % (kind, l), 0.1, 18.0 * cin * cout * (bench.TILE // scale[l]) ** 2))
            return rows + [('gram conv1_1', 0.05, 8.6e9), ('sums', 0.01, 0.0)]

    class Job:
        eng = Eng()
    rec = bench.kernel_records(Job(), reps=2)
    groups = {g['name']: g for g in rec['groups']}
    fwd = next(g for n, g in groups.items() if n.startswith('forward 3x3 layers from 128'))
    bwd = next(g for n, g in groups.items() if n.startswith('backward 3x3 layers from 128'))
    inj = next(g for n, g in groups.items() if n.startswith('loss-injecting backward'))
    assert (fwd['launch_groups'], bwd['launch_groups'], inj['launch_groups']) == (10, 6, 4)
    direct = sum(18.0 * chans[l][0] * chans[l][1] * (1024 // scale[l]) ** 2 for l in convs
                 if min(chans[l]) >= 128)
    assert fwd['flop_issued_fp16'] == pytest.approx(2 * direct)
    assert fwd['frac_fp16_pipe'] == pytest.approx(2 * direct / 1e-3 / 1e12 / (16 * 157.3), rel=1e-6)
    assert rec['dominant'] is fwd and groups['loss terms, pooling, copies']['launch_groups'] == 2
    shallow = [n for n in groups if n.startswith('64-channel layers')]
    assert sorted(n.split(': ')[1].split(' (')[0] for n in shallow) == ['bwd conv1_2', 'bwd conv2_1', 'fwd conv1_2', 'fwd conv2_1']
    assert groups['first layer / backward into the image (fp32 MFMA)']['launch_groups'] == 2
    assert rec['tile_ms_single_stream'] == pytest.approx(0.1 * 2 * len(convs) + 0.06)
