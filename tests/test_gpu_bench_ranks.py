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


def test_farm_leg_over_two_device_entries_matches_one():
    """bench.py's `farm` sub-record (north_star's layout: one host process, TileFarm over the
    job's GPUs) on a one-GPU box: the device list [0, 0] gives two groups of four engines on GPU 0,
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
            assert len(job.farm.engines) == 8 and len(job.farm.primaries()) == 1
            assert job.eng.query(lib.Q_SHARED_ENGINES) == 8
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
