"""The one-process-per-GPU protocol (scatter tiles / evaluate / gather gradients) with
world_size 2 over gloo on CPU.  The arithmetic plugged in is the numpy oracle; the check is that
the distributed stitch equals the single-process evaluation bit for bit."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.num_ops import roll_xy
from oracle.tile_path import tile_grid
from tests.helpers import DEFAULT_STYLE_LAYERS, make_oracle, normalized_weights


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _problem():
    rng = np.random.RandomState(11)
    img = rng.uniform(-110, 120, (3, 48, 72)).astype(np.float32)
    style = rng.uniform(-110, 120, (3, 40, 44)).astype(np.float32)
    cl, cw = normalized_weights(['conv4_2'], 0.05)
    sl, sw = normalized_weights(DEFAULT_STYLE_LAYERS, 1)
    return img, style, cl, cw, sl, sw


def _worker(rank, world, port, out_path, two_phase=False, tile_size=32):
    from style_transfer_amd.dist import DistributedTiles, broadcast_targets, broadcast_weights
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    img, style, cl, cw, sl, sw = _problem()
    om, _ = make_oracle('vgg16_avgpool')
    contents, styles = [], []
    if rank == 0:
        styles = [om.style_grams([style], sl, 512)]
        contents = [om.prepare_features(img, cl, 512)]
    contents, styles = broadcast_targets(contents, styles, 'cpu')
    om.contents = [{k: v.numpy() for k, v in c.items()} for c in contents]
    om.styles = [{k: v.numpy() for k, v in s_.items()} for s_ in styles]
    # the filter bank travels once, as one broadcast: rank 1 starts without it
    bank = broadcast_weights(om.net.params if rank == 0 else None, 'cpu')
    for name, (w, b) in make_oracle('vgg16_avgpool')[0].net.params.items():
        assert np.array_equal(bank[name][0].numpy(), w) and np.array_equal(bank[name][1].numpy(), b)
    roll = (16, -8)
    rolled = roll_xy(img.copy(), roll)
    grad = np.zeros_like(img)

    def cut(rect, r):
        return torch.from_numpy(np.ascontiguousarray(rolled[:, rect[0]:rect[1], rect[2]:rect[3]]))

    def evaluate(jobs, r):
        out = []
        om.roll_contents(r)
        for tile, start in jobs:
            loss, g = om.sc_grad_tile(tile.numpy(), start, cl, sl, {}, cw, sw)
            out.append((loss, torch.from_numpy(g)))
        om.roll_contents((-r[0], -r[1]))
        return out

    def put(rect, g, r):
        grad[:, rect[0]:rect[1], rect[2]:rect[3]] = g.numpy()

    pending = {}

    def begin(jobs, r):              # the two-phase form: rank 0 overlaps its sends with this
        pending['args'] = (jobs, r)

    def end():
        return evaluate(*pending.pop('args'))

    farm = DistributedTiles(cut, (begin, end) if two_phase else evaluate, put, 'cpu')
    rects = tile_grid(img.shape[-2:], tile_size)
    # only rank 0 knows the shift; it reaches rank 1 with the tiles, not through a collective
    loss = farm.eval_sc_grad(rects, roll if rank == 0 else None)
    if rank == 0:
        np.savez(out_path, loss=loss, grad=grad, n_tiles=len(rects))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('two_phase', [False, True])
def test_two_rank_tile_farm_equals_single_process(tmp_path, two_phase):
    out = str(tmp_path / 'dist.npz')
    mp.spawn(_worker, args=(2, _free_port(), out, two_phase), nprocs=2, join=True)
    got = np.load(out)
    img, style, cl, cw, sl, sw = _problem()
    om, _ = make_oracle('vgg16_avgpool')
    om.styles = [om.style_grams([style], sl, 512)]
    om.contents = [om.prepare_features(img, cl, 512)]
    roll = (16, -8)
    ref_loss, ref_grad = om.sc_grad(roll_xy(img.copy(), roll), roll, 32, cl, sl, {}, cw, sw)
    assert int(got['n_tiles']) == 6
    assert float(got['loss']) == pytest.approx(ref_loss, rel=1e-12)
    assert np.array_equal(got['grad'], ref_grad)


@pytest.mark.timeout(600)
def test_strong_layout_with_an_idle_rank_equals_single_process(tmp_path):
    """bench.py's strong layout at N = 8 has more ranks than tiles (four tiles of the fixed
    2048 x 2048 image, eight GPUs): ranks without a tile take part in nothing but the barriers.
    Here: two tiles, three ranks -- rank 2 owns no tile, receives no header and sends nothing; the
    stitched result equals the single-process evaluation bit for bit and the loss is added up in
    tile order."""
    out = str(tmp_path / 'dist3.npz')
    mp.spawn(_worker, args=(3, _free_port(), out, True, 48), nprocs=3, join=True)
    got = np.load(out)
    img, style, cl, cw, sl, sw = _problem()
    om, _ = make_oracle('vgg16_avgpool')
    om.styles = [om.style_grams([style], sl, 512)]
    om.contents = [om.prepare_features(img, cl, 512)]
    roll = (16, -8)
    ref_loss, ref_grad = om.sc_grad(roll_xy(img.copy(), roll), roll, 48, cl, sl, {}, cw, sw)
    assert int(got['n_tiles']) == 2
    assert float(got['loss']) == ref_loss
    assert np.array_equal(got['grad'], ref_grad)
