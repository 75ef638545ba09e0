"""bench.py's one-process-per-GPU protocol on a ONE-GPU box: two ranks share GPU 0
(STX_BENCH_DEBUG_ONE_GPU=1; tiles travel over gloo through host memory instead of RCCL / xGMI).
Same code path as `--gpus 2` otherwise: rank 0 owns the image and the optimizer, the weights
arrive by broadcast, the shift travels with the tiles.  The headline is BASELINE's literal
workload (the fixed 2048 x 2048 image, four tiles per step: strong scaling, two tiles per rank);
`weak` (2048 x 4096, four tiles per rank), `farm` (the same 2048 x 2048 workload through ONE host
process, TileFarm with the cross-GPU staging leg forced) and `config4` (4096 x 4096, 16 tiles,
L-BFGS) are sub-records, and each carries `bit_identical`: step 1 evaluated by all ranks / all
device entries against GPU 0 alone.  The tiles of a step are independent, so the loss after a
few steps must not depend on how many ranks shared the work."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _line(stdout):
    rows = [l for l in stdout.splitlines() if l.startswith('{')]
    assert rows, stdout[-2000:]
    return json.loads(rows[-1])


@pytest.mark.timeout(900)
def test_two_rank_bench_matches_single_process_loss():
    env = dict(os.environ, STX_BENCH_DEBUG_ONE_GPU='1', MASTER_ADDR='127.0.0.1')
    common = ['--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-wall-clock',
              '--steady-seconds', '0']
    for attempt in range(2):                # (a second try with another port: the rendezvous, not the bench, is what can fail)
        two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                              '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                              str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', '2'] + common,
                             env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             text=True, timeout=800)
        if two.returncode == 0:
            break
        print('attempt %d failed:\n%s' % (attempt, two.stdout[-3000:]))
    assert two.returncode == 0, two.stdout[-3000:]
    a = _line(two.stdout)
    assert a['n_gpus'] == 2 and a['scaling'] == 'strong'
    assert a['config']['tiles_per_step'] == 4 and a['config']['tiles_per_gpu'] == 2
    assert a['config']['idle_gpus'] == 0 and '2048x2048' in a['config']['workload']
    assert a['bit_identical'] is True
    assert a['weak']['bit_identical'] is True and '4096x2048' in a['weak']['workload']
    for key, tiles in (('farm', 4), ('config4', 16)):
        rec = a[key]
        assert 'error' not in rec, rec
        assert rec['bit_identical'] is True and rec['scaling'] == 'strong' and rec['value'] > 0
        assert '%d tiles' % tiles in rec['workload']
    assert 'lbfgs' in a['config4']['workload'] and '4096x4096' in a['config4']['workload']
    one = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1'] + common,
                         env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=800)
    assert one.returncode == 0, one.stdout[-3000:]
    b = _line(one.stdout)
    assert b['config']['tiles_per_step'] == 4 and b['scaling'] == 'strong'
    # same tiles, same arithmetic, losses added up in tile order in both layouts: the same bits
    assert a['config']['final_loss'] == b['config']['final_loss']
    assert a['farm']['final_loss'] == b['config']['final_loss']
    # the N = 1 line explains itself: the fraction stated against the fp16 pipe, the dominant kernel measured
    # in the run, the strict-fp32 figure timed beside the headline
    roof = b['roofline']
    assert roof['peak_fp16'] == pytest.approx(16 * roof['peak']) and roof['frac_fp16_pipe'] == pytest.approx(roof['frac'])
    dom = roof['dominant']
    assert 'conv_h2_kernel<0,2,1,*>' in dom['name'] and dom['launch_groups'] == 10
    assert 0.1 < dom['frac_fp16_pipe'] < 1 and 0.2 < dom['share_of_tile'] < 0.5
    assert 0 < roof['per_kernel']['furthest_below']['frac_fp16_pipe'] <= dom['frac_fp16_pipe']
    f32 = b['fp32_kernels']
    assert f32['dtype'] == 'f32' and 0 < f32['value'] < b['value']
    assert f32['final_loss'] == pytest.approx(b['config']['final_loss'], rel=1e-4)
    assert f32['conv_flop_issued_over_direct'] == pytest.approx(4 / 9, rel=0.05)      # (fp32 2-D Winograd: 4 / 9 of a direct count)


def test_farm_leg_over_two_device_entries_matches_one():
    """bench.py's `farm` sub-record (north_star's layout: one host process, TileFarm over the
    job's GPUs) on a one-GPU box: the device list [0, 0] gives two groups of STREAMS_PER_GPU engines on GPU 0,
    eight tiles per step.  Same tiles, same arithmetic: the loss after two steps equals the
    single-entry farm's bit for bit, and the second device entry shares the first one's weights
    and targets (one copy per GPU)."""
    sys.path.insert(0, REPO)
    import bench
    from style_transfer_amd import lib
    from style_transfer_amd.netspec import builtin_net
    from style_transfer_amd.weights import synthetic_weights
    if lib.device_count() < 1:
        pytest.fail('no GPU visible')
    net = builtin_net('vgg19')
    weights = synthetic_weights(net, 0)
    losses = []
    for devices in ([0], [0, 0]):
        job = bench.FarmJob(net, weights, devices, 2, 4)
        if len(devices) == 2:
            assert job.bit_identical()
        _, loss = job.timed(2, 1)
        assert job.timed_tile_evals == 2 * 8 and len(job.group_ms) == 2
        if len(devices) == 2:
            n_eng = 2 * bench.STREAMS_PER_GPU
            assert len(job.farm.engines) == n_eng and len(job.farm.primaries()) == 1
            assert job.eng.query(lib.Q_SHARED_ENGINES) == n_eng
        losses.append(loss)
        job.close()
    assert losses[0] == losses[1]


def test_farm_leg_child_process_prints_its_record():
    """`bench.py --farm-leg N` (what rank 0 spawns for the `farm` sub-record at N > 1) on one GPU."""
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--farm-leg', '0',
                           '--debug-grid', '2x2', '--steps', '2', '--warmup', '1'], cwd=REPO,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    rec = _line(proc.stdout)
    assert rec['unit'] == 'tile-iterations/s' and rec['value'] > 0 and rec['steps'] == 2
    assert rec['tile_evals'] == 4 * 3 + 8 and rec['bit_identical'] is True     # (+ the identity check's two evaluations)


def _single_rank_rccl(out_path):
    """Child process: a one-rank RCCL group on GPU 0 next to libstx in the same process."""
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from style_transfer_amd.dist import DistributedTiles, broadcast_targets, broadcast_weights
    from style_transfer_amd.engine import TileEngine
    from style_transfer_amd.netspec import builtin_net
    from style_transfer_amd.weights import synthetic_weights
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    net = builtin_net('vgg19')
    host_bank = synthetic_weights(net, 0)
    bank = broadcast_weights(host_bank, device)
    eng = TileEngine(net, 0, bank)
    rng = np.random.RandomState(0)
    cmap = eng.to_device(np.abs(rng.standard_normal((512, 16, 24))).astype(np.float32))
    gram = np.tril(rng.standard_normal((64, 64))).astype(np.float32)
    contents, styles = broadcast_targets([{'conv4_2': cmap}], [{'conv1_1': gram}], device)
    t = contents[0]['conv4_2']
    result = {'in_place': bool(t.is_cuda and t.data_ptr() == cmap.ptr),
              'content_equal': bool(np.array_equal(t.cpu().numpy(), cmap.get())),
              'gram_equal': bool(np.array_equal(styles[0]['conv1_1'].cpu().numpy(), gram)),
              'weights_equal': all(bool(np.array_equal(bank[k][0].cpu().numpy(), np.asarray(host_bank[k][0])))
                                   for k in host_bank)}
    # the tile protocol with a single rank: every tile is local, nothing is sent
    calls = []
    tiles = DistributedTiles(lambda rect, roll: torch.zeros((3, rect[1] - rect[0], rect[3] - rect[2]), device=device),
                             lambda jobs, roll: [(1.5, tile) for tile, _ in jobs],
                             lambda rect, g, roll: calls.append(rect), device)
    result['loss'] = tiles.eval_sc_grad([(0, 8, 0, 8), (0, 8, 8, 16)], (8, 16))
    result['puts'] = len(calls)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()
    with open(out_path, 'w') as f:
        json.dump(result, f)


def test_single_rank_rccl_group_broadcasts_device_targets_in_place(tmp_path):
    """RCCL (torch.distributed backend 'nccl') initialises next to libstx in one process, and
    dist.broadcast_targets hands a DeviceArray to the collective WHERE IT LIES: the tensor that
    goes on the wire has the DeviceArray's own address (no bounce through host memory -- a
    537 MB content map at a 4096 x 4096 scale).  One rank is what a one-GPU box can run; the
    multi-rank protocol is covered on CPU by tests/test_dist_gloo.py."""
    out = str(tmp_path / 'rccl.json')
    code = 'import sys; sys.path.insert(0, %r); from tests.test_gpu_bench_ranks import _single_rank_rccl; ' \
           '_single_rank_rccl(%r)' % (REPO, out)
    proc = subprocess.run([sys.executable, '-c', code], cwd=REPO, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    got = json.load(open(out))
    assert got == {'in_place': True, 'content_equal': True, 'gram_equal': True, 'weights_equal': True,
                   'loss': 3.0, 'puts': 2}, got
