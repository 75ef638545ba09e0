"""bench.py's one-process-per-GPU protocol on a ONE-GPU box: two ranks share GPU 0
(STX_BENCH_DEBUG_ONE_GPU=1; tiles travel over gloo through host memory instead of RCCL / xGMI).
Same code path as `--gpus 2` otherwise: rank 0 owns the image and the optimizer, the weights
arrive by broadcast, the shift travels with the tiles, both ranks evaluate four tiles per step.
The tiles of a step are independent, so the loss after a few steps must not depend on how many
ranks shared the work: the 2-rank run on a 2048 x 4096 image is compared with the same image
evaluated by a single process."""

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
    assert a['n_gpus'] == 2 and a['config']['tiles_per_step'] == 8
    one = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1',
                          '--debug-grid', '2x4'] + common, env=env, cwd=REPO,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800)
    assert one.returncode == 0, one.stdout[-3000:]
    b = _line(one.stdout)
    assert b['config']['tiles_per_step'] == 8
    assert a['config']['final_loss'] == pytest.approx(b['config']['final_loss'], rel=1e-6)


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
        _, loss = job.timed(2, 1)
        assert job.farm.tile_evals == 3 * 8 and len(job.group_ms) == 2
        if len(devices) == 2:
            assert len(job.farm.engines) == 8 and len(job.farm.primaries()) == 1
            assert job.eng.query(lib.Q_SHARED_ENGINES) == 8
        losses.append(loss)
        job.close()
    assert losses[0] == losses[1]


def test_farm_leg_child_process_prints_its_record():
    """`bench.py --farm-leg N` (what rank 0 spawns for the `farm` sub-record at N > 1) on one GPU."""
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--farm-leg', '1',
                           '--debug-grid', '2x2', '--steps', '2', '--warmup', '1'], cwd=REPO,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    rec = _line(proc.stdout)
    assert rec['unit'] == 'tile-iterations/s' and rec['value'] > 0 and rec['steps'] == 2
    assert rec['graphs']['eager'] == 4 * 3
