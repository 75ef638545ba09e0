#!/usr/bin/env python3
"""Headline benchmark: tile-iterations/s of the tiled style-transfer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one optimizer iteration of the reference's step loop (style_transfer.py:771-806) at the
top scale of the configuration BASELINE.json's metric is quoted on ("VGG-19 --size 2048 --tile-size
1024, Adam"): draw the seam-suppression shift, cut the four 1024 x 1024 tiles, per-tile VGG-19
forward + Gram/content losses + backward (stx_sc_grad_tile; the four tiles run concurrently on four
HIP streams of the GPU), stitch, TV + p-norm regularizers, fused Adam step with iterate averaging,
step statistics.  Everything is resident in HBM when the timed region starts.  Synthetic data:
seeded He-initialised VGG-19 weights, seeded low-pass-noise content and style pictures (no network).

At N = 1 the step loop runs through TileFarm, the product's own driver.  N > 1 is weak scaling:
every rank evaluates four 1024 x 1024 tiles per step and the image grows with N (2048 x 4096,
4096 x 4096 -- config 4's top scale -- and 4096 x 8192 at N = 2, 4, 8); rank 0 owns the image and
the optimizer, tiles go out and gradients come back as batched point-to-point transfers over RCCL,
there is no collective on the data path.  value = tile-iterations per second of the whole job.
north_star's layout -- ONE host process driving all N GPUs through TileFarm (tiles and gradients
as xGMI peer copies ordered by events, no host wait inside a step) -- is timed on the same image
and step loop after the ranks have finished, and reported as the `farm` sub-record.

The line also carries
  roofline      the time the kernels' own instructions need on the matrix pipe, in fp32-MFMA FLOP,
                for the four tile-iterations of one GPU (convolutions through Winograd F(2x2,3x3)
                issue 4/9 of a direct convolution's MFMAs; Gram and SYMM run as six bf16 MFMAs per
                16 k and are counted at the 0.375 of their fp32 pipe time that this occupies) over
                the GPU time of that concurrent group of stx_sc_grad_tile calls -- HIP events on
                each engine's own stream inside the timed region, the longest of the four spans --
                against the fp32 MFMA peak, so that frac <= 1 by construction; `bound_ms` is the
                time the same work takes at that peak; `frac_round2_accounting` counts Gram and
                SYMM in full, as round 2 did.  The SURVEY 8d figure (1 514 240 FLOP per tile pixel,
                every convolution counted as a direct one) is kept beside it as
                achieved_direct_equiv;
  steady        the same step loop run for at least 5 s after the timed region;
  wall_clock_s  the WHOLE `--size 2048 --tile-size 1024` command-line run (7 pyramid scales,
                800 iterations, 1400 tile-iterations, preprocessing and PNG output included) on
                synthetic pictures, on this job's GPUs through one host process (TileFarm);
  cpu_baseline  the numpy oracle (a port of the reference's Caffe-CPU path) timed on this box's
                host cores on one 1024 x 1024 tile-iteration -- rank 0, N = 1 only.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

TILE = 1024
FLOP_PER_TILE_PIXEL = 1514240          # VGG-19, default taps: fwd + dgrad + Gram + SYMM
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 2.4 GHz
NOMINAL_CLOCK_MHZ = 2400.0
TILES_PER_GPU = 4
GRIDS = {1: (2, 2), 2: (2, 4), 4: (4, 4), 8: (4, 8)}      # tile grid of the image per world size
CONTENT_LAYERS = ['conv4_2']
STYLE_LAYERS = ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
MEAN = (103.939, 116.779, 123.68)


def smooth_picture(seed, h, w):
    """Low-pass filtered seeded noise, float32 BGR minus mean, [3,h,w]."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    small = rng.uniform(0, 255, (max(2, h // 16), max(2, w // 16), 3)).astype(np.uint8)
    big = np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC), np.float32)
    big = np.clip(big + rng.uniform(-16, 16, big.shape), 0, 255)
    return np.ascontiguousarray(big.transpose(2, 0, 1)[::-1] - np.float32(MEAN).reshape(3, 1, 1))


TRAFFIC_PROFILE = 'profiles/r03_hbm_traffic_pmc.json'


def measured_traffic():
    """(bytes per launch group, source file): HBM bytes from the PMC passes over THIS script
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per
    the gfx950 calibration on the Adam kernel -- tools/pmc_traffic.py).  Counters cannot be
    collected inside a timed run, so the figure is a replay of the committed profile of this
    build state and the line names the file; (None, None) if it is missing."""
    try:
        with open(os.path.join(REPO, TRAFFIC_PROFILE)) as f:
            # measured per tile-iteration; one launch group = 4 of them
            return float(json.load(f)['hbm_bytes_per_tile_iteration']) * TILES_PER_GPU, TRAFFIC_PROFILE
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline(net):
    """Times the oracle's tile evaluation on the host cores (checker used as a yardstick only)."""
    from oracle.caffe_net import synthetic_weights
    from oracle.tile_path import OracleModel
    size = TILE
    layers = net.as_dicts()
    om = OracleModel(layers, synthetic_weights(layers, 0))
    rng = np.random.RandomState(1)
    tile = smooth_picture(2, size, size)
    cw = {'conv4_2': 0.05}
    sw = {l: 0.2 for l in STYLE_LAYERS}
    om.contents = [{'conv4_2': np.abs(rng.standard_normal((512, size // 8, size // 8))).astype(np.float32)}]
    om.styles = [{l: np.tril(rng.standard_normal((om.channels[l],) * 2)).astype(np.float32)
                  for l in STYLE_LAYERS}]
    om.sc_grad_tile(tile[:, :128, :128], (0, 0), CONTENT_LAYERS, STYLE_LAYERS, {}, cw, sw)  # warm
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or (time.perf_counter() - t0 < 12 and reps < 3):
        om.sc_grad_tile(tile, (0, 0), CONTENT_LAYERS, STYLE_LAYERS, {}, cw, sw)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    return {'value': 1.0 / dt, 'unit': 'tile-iterations/s', 'cores': cores, 'kind': 'port',
            'sample': '%d VGG-19 tile-iteration(s) at the benchmark tile size %dx%d with the numpy '
                      'oracle (im2col + multithreaded SGEMM, %.2f s each), no scaling'
                      % (reps, size, size, dt)}


class FarmJob:
    """The benchmark's step loop on a TileFarm over `devices` (one host process): image of
    rows x cols tiles of 1024 x 1024, targets computed on the master GPU, Adam."""

    def __init__(self, net, weights, devices, rows, cols):
        from style_transfer_amd import image_ops
        from style_transfer_amd.farm import TileFarm
        from style_transfer_amd.optimizers import AdamOptimizer
        self.image_ops = image_ops
        self.H, self.W = rows * TILE, cols * TILE
        self.farm = TileFarm(net, list(devices), weights, verbose=False,
                             streams_per_device=TILES_PER_GPU)
        eng = self.eng = self.farm.master
        style_feats = self.farm.prepare_features_device(smooth_picture(7, TILE, TILE), STYLE_LAYERS,
                                                        TILE, passes=1)
        styles = [{l: eng.gram_matrix(f) for l, f in style_feats.items()}]
        contents = [self.farm.prepare_features_device(smooth_picture(8, self.H, self.W),
                                                      CONTENT_LAYERS, TILE, passes=1)]
        self.farm.set_contents_and_styles(contents, styles)
        self.content_weight = {'conv4_2': 0.05}
        self.style_weight = {l: 1.0 / len(STYLE_LAYERS) for l in STYLE_LAYERS}
        rng = np.random.RandomState(0)
        # the reference's start image: uniform noise minus the mean (style_transfer.py:889)
        self.img = eng.to_device(rng.uniform(0, 255, (3, self.H, self.W)).astype(np.float32) -
                                 np.float32(MEAN).reshape(3, 1, 1))
        self.grad = eng.empty((3, self.H, self.W))
        self.old = eng.empty((3, self.H, self.W)).copy_from(self.img)
        self.rng = np.random.RandomState(0)
        self.opt = AdamOptimizer(eng, self.img, step_size=15, bp1=1 - 1 / 20, decay=0.05, power=0.5)
        self.group_ms = []
        self.clock = None
        self.tiles_per_step = rows * cols

    def step(self):
        """One iteration of the reference's step loop (style_transfer.py:771-815); one host
        synchronisation, for the step statistics."""
        xy = np.int32(self.rng.uniform(-0.5, 0.5, size=2) * (self.H, self.W)) // 8
        roll = xy * 8

        def opfunc(params):
            loss = self.farm.eval_sc_grad(params, self.grad, roll, CONTENT_LAYERS, STYLE_LAYERS, {},
                                          self.content_weight, self.style_weight, TILE, lazy=True)
            loss.add(self.image_ops.regularizers(self.eng, params, self.grad, MEAN, 5.0, 2.0, 2.0,
                                                 6.0), self.eng)
            return loss, self.grad
        avg, loss = self.opt.update(opfunc)
        self.image_ops.step_stats(self.eng, avg, self.old)
        loss = float(loss)
        self.group_ms.append(max(e.last_tile_ms() for e in self.farm.engines[:self.tiles_per_step]))
        return loss

    def fence(self):
        for e in self.farm.engines:
            e.sync()

    def timed(self, steps, warmup, clock_marks=False):
        for _ in range(warmup):
            self.step()
        self.fence()
        self.group_ms.clear()
        # (clock_marks: the engines record the shader clock twice per tile evaluation, 20 us each --
        # the second, longer measurement only, never the headline's timed steps)
        engines = self.farm.engines[:self.tiles_per_step] if clock_marks else []
        for e in engines:
            e.clock_marks(True)
        t0 = time.perf_counter()
        loss = None
        for _ in range(steps):
            loss = self.step()
        self.fence()
        elapsed = time.perf_counter() - t0
        if clock_marks:
            self.clock = clock_summary(engines)
        return elapsed, loss

    def graph_counters(self):
        from style_transfer_amd import lib
        return {'captures': sum(e.query(lib.Q_GRAPH_CAPTURES) for e in self.farm.engines),
                'replays': sum(e.query(lib.Q_GRAPH_REPLAYS) for e in self.farm.engines),
                'eager': sum(e.query(lib.Q_EAGER_TILES) for e in self.farm.engines)}

    def close(self):
        self.farm.close()


def farm_leg(devices, rows, cols, steps, warmup):
    """The benchmark's step loop through TileFarm over `devices` in THIS process; the `farm`
    sub-record."""
    from style_transfer_amd.netspec import builtin_net
    from style_transfer_amd.weights import synthetic_weights
    net = builtin_net('vgg19')
    job = FarmJob(net, synthetic_weights(net, 0), devices, rows, cols)
    elapsed, loss = job.timed(steps, warmup)
    record = {'layout': 'one host process, TileFarm over %d GPUs (xGMI peer copies, event-ordered, '
                        'no host wait inside a step)' % len(devices),
              'value': job.tiles_per_step * steps / elapsed, 'unit': 'tile-iterations/s',
              'ms_per_step': elapsed / steps * 1e3, 'steps': steps, 'final_loss': loss,
              'avg_launch_ms': float(np.mean(job.group_ms)), 'graphs': job.graph_counters()}
    job.close()
    return record


def farm_leg_in_child(world, rows, cols, steps, warmup, timeout=300):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--farm-leg', str(world), '--debug-grid',
           '%dx%d' % (rows, cols), '--steps', str(steps), '--warmup', str(warmup)]
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK')}
    try:
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              timeout=timeout)
        rows_ = [l for l in proc.stdout.splitlines() if l.startswith('{')]
        if proc.returncode != 0 or not rows_:
            return {'error': 'exit %d: %s' % (proc.returncode, proc.stderr.strip()[-400:])}
        return json.loads(rows_[-1])
    except Exception as err:      # pylint: disable=broad-except
        return {'error': '%s: %s' % (type(err).__name__, err)}


def whole_run_wall_clock(devices):
    """Wall-clock of the reference's command line for the metric's configuration, start to finish:
    `--size 2048 --tile-size 1024`, Adam, default iterations (200 + 6 x 100 over 7 scales = 1400
    tile-iterations), synthetic 2048 x 2048 pictures, seeded synthetic weights, ONE host process
    driving `devices` through TileFarm.  Runs as a child process (a failure there cannot take the
    benchmark line with it).  wall_clock_s is what the command itself reports in its last line,
    like the reference (style_transfer.py:1152-1163); process_wall_s includes interpreter and HIP
    start-up."""
    import re
    import subprocess
    import tempfile
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix='stx_bench_')

    def picture(seed, name):
        r = np.random.RandomState(seed)
        small = r.uniform(0, 255, (128, 128, 3)).astype(np.uint8)
        big = np.asarray(Image.fromarray(small).resize((2048, 2048), Image.BICUBIC), np.float32)
        Image.fromarray(np.uint8(np.clip(big + r.uniform(-16, 16, big.shape), 0, 255))).save(
            os.path.join(tmp, name))
    picture(0, 'content.png')
    picture(1, 'style.png')
    args = ['-ci', 'content.png', '-si', 'style.png', '--size', '2048', '--tile-size', '1024',
            '--weights', 'synthetic', '--display', 'none', '-oi', 'out.png',
            '--devices'] + [str(d) for d in devices]
    t0 = time.perf_counter()
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'style_transfer.py')] + args, cwd=tmp,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    outer = time.perf_counter() - t0
    lines = proc.stdout.splitlines()
    if proc.returncode != 0:
        raise RuntimeError('style_transfer.py exited with %d: %s' % (proc.returncode,
                                                                     ' | '.join(lines[-3:])))
    m = re.search(r'ending after (\d+)m ([\d.]+)s', proc.stdout)
    wall = int(m.group(1)) * 60 + float(m.group(2)) if m else outer
    summary = [l for l in lines if 'tile-iterations in' in l or 'ending after' in l]
    return {'wall_clock_s': wall, 'process_wall_s': outer,
            'wall_clock_command': 'style_transfer.py ' + ' '.join(args),
            'wall_clock_steps': sum(1 for l in lines if l.startswith('Step ')),
            'wall_clock_summary': summary}


# Gram + SYMM of the five default style layers of VGG-19: each product is 2 * C * C * h * w FLOP and
# C * C * h * w = 2^32 / 2^20 per tile pixel for conv1_1 .. conv4_1 (a quarter of that for conv5_1)
GRAM_SYMM_FLOP_PER_TILE_PIXEL = int(2 * 2 * 4096 * (4 + 0.25))
BF16X3_PIPE_TIME = 6 * 32 / (8 * 64)    # six bf16 MFMAs (32 cycles) per 16 k instead of eight fp32 ones (64)


def clock_summary(engines):
    """Reads and switches off the engines' clock marks: {'mhz': median, 'p10', 'p90', 'min', 'max',
    'samples'} or None."""
    mhz = []
    for e in engines:
        mhz += [m for m in e.clock_marks_read() if m > 0]
        e.clock_marks(False)
    if not mhz:
        return None
    return {'mhz': float(np.median(mhz)), 'p10': float(np.percentile(mhz, 10)),
            'p90': float(np.percentile(mhz, 90)), 'min': float(np.min(mhz)), 'max': float(np.max(mhz)),
            'samples': len(mhz)}


def add_clock(roofline, clock):
    """roofline.clock_mhz (+ percentiles), peak_at_clock, frac_at_clock from a clock_summary."""
    if not clock:
        return
    # (a reading is good to about 3 %: the median may come out above the part's 2.4 GHz; never credit more)
    peak_at_clock = PEAK_FP32_MFMA_TFLOPS * min(clock['mhz'], NOMINAL_CLOCK_MHZ) / NOMINAL_CLOCK_MHZ
    roofline.update({'clock_mhz': clock['mhz'], 'clock_mhz_p10': clock['p10'],
                     'clock_mhz_p90': clock['p90'], 'clock_mhz_min': clock['min'],
                     'clock_mhz_max': clock['max'], 'clock_samples': clock['samples'],
                     'peak_at_clock': peak_at_clock,
                     'frac_at_clock': roofline['achieved'] / peak_at_clock})


def roofline_record(eng, avg_group_ms):
    """Matrix-pipe work of one GPU's four concurrent tile evaluations over their HIP-event span,
    against the fp32 MFMA peak.  Every term is the time the kernels' own instructions need on the
    matrix pipe, expressed in fp32-MFMA FLOP: Winograd convolutions issue 4/9 (2-D) or 2/3 (1-D) of
    a direct convolution's fp32 MFMAs; Gram and SYMM issue six bf16 MFMAs per 16 k, which occupy
    the pipe for 0.375 of the time their fp32 form would -- so frac <= 1 by construction."""
    flop = FLOP_PER_TILE_PIXEL * TILE * TILE * TILES_PER_GPU
    direct_equiv = flop / (avg_group_ms * 1e-3) / 1e12
    conv_alg, conv_issued = eng.last_tile_flops()
    terms = GRAM_SYMM_FLOP_PER_TILE_PIXEL * TILE * TILE
    bf16_terms = os.environ.get('STX_GRAM') != 'fp32' and os.environ.get('STX_SYMM') != 'fp32'
    per_tile = flop / TILES_PER_GPU - conv_alg - terms + conv_issued
    issued_r02 = (per_tile + terms) * TILES_PER_GPU            # round-2 accounting: terms at the fp32 rate
    issued = (per_tile + terms * (BF16X3_PIPE_TIME if bf16_terms else 1.0)) * TILES_PER_GPU
    issued_tflops = issued / (avg_group_ms * 1e-3) / 1e12
    bound_ms = issued / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
    traffic, traffic_src = measured_traffic()
    return {'bound': 'mfma', 'achieved': issued_tflops, 'peak': PEAK_FP32_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': issued_tflops / PEAK_FP32_MFMA_TFLOPS,
            'frac_round2_accounting': issued_r02 / (avg_group_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'bound_ms': bound_ms, 'bound_ms_per_tile': bound_ms / TILES_PER_GPU,
            'traffic': traffic, 'traffic_unit': 'bytes per launch',
            'traffic_source': traffic_src,
            'kernel': 'stx_sc_grad_tile x %d concurrent on one GPU: conv_wino2_kernel<0|1|3,32> '
                      '(3x3 layers forward / backward / loss-injecting backward; 80 %% of the time), '
                      'conv_mfma_kernel (first layer), conv3x3_m4_kernel (backward into the image), '
                      'gram_partial_bf3_kernel / symm_bf3_kernel (bf16 MFMA, three-piece split)'
                      % TILES_PER_GPU,
            'flop_issued_per_launch': issued, 'avg_launch_ms': avg_group_ms,
            'achieved_direct_equiv': direct_equiv,
            'flop_direct_equiv_per_launch': flop,
            'note': 'achieved / frac = matrix-pipe work actually issued, in fp32-MFMA FLOP (Winograd '
                    'F(2x2,3x3) convolutions issue 4/9 of a direct convolution; Gram and SYMM run as '
                    'six bf16 MFMAs per 16 k = 0.375 of the pipe time of their fp32 form and are '
                    'counted at that) over the HIP-event time of the launch group, against the fp32 '
                    'MFMA peak at 2.4 GHz; clock_mhz is the shader clock during the second, longer measurement '
                    '(`steady`; stx_clock_marks: two 20-microsecond readings of core cycles against the '
                    '100 MHz counter per tile evaluation, each good to about 3 %; the median), peak_at_clock '
                    'the fp32 MFMA peak at min(that clock, 2.4 GHz) and frac_at_clock = achieved / '
                    'peak_at_clock; '
                    'frac_round2_accounting counts Gram and SYMM in full as '
                    'round 2 did (they ran on the fp32 pipe then); achieved_direct_equiv credits '
                    'every convolution as a direct one (SURVEY 8d: 1 514 240 FLOP per tile pixel) '
                    'and is not a roofline fraction; traffic is a replay of the committed PMC '
                    'profile named in traffic_source, not a measurement of this run'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-wall-clock', action='store_true',
                    help='skip the whole-run wall-clock leg (about 10 s)')
    ap.add_argument('--no-farm-leg', action='store_true',
                    help='N > 1: skip the single-host-process (TileFarm) measurement')
    ap.add_argument('--farm-leg', type=int, default=0, metavar='N',
                    help='internal: only the single-host-process (TileFarm) step loop over GPUs 0..N-1 '
                         'with the tile grid of --debug-grid; prints the `farm` sub-record')
    ap.add_argument('--steady-seconds', type=float, default=5.0)
    ap.add_argument('--debug-grid', default=None,
                    help='RxC tile grid instead of the one for --gpus (tests only: lets a single '
                         'process evaluate the image of a larger job)')
    opts = ap.parse_args()

    if opts.farm_leg:
        r, c = (int(v) for v in opts.debug_grid.split('x'))
        print(json.dumps(farm_leg(list(range(opts.farm_leg)), r, c, opts.steps, opts.warmup)), flush=True)
        return
    import torch                                       # first: one HIP runtime for both libraries
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != opts.gpus:
        if world == 1 and opts.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run '
                     '(one process per GPU)' % opts.gpus)
        sys.exit('WORLD_SIZE=%d does not match --gpus %d' % (world, opts.gpus))
    if opts.gpus not in GRIDS:
        sys.exit('--gpus must be one of %s' % sorted(GRIDS))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an AMD GPU (no CPU path exists)')
    # STX_BENCH_DEBUG_ONE_GPU=1 (debugging the N > 1 protocol on a 1-GPU box only): every rank
    # uses GPU 0 and tiles travel over gloo through host memory.  Never a benchmark number.
    debug_one_gpu = os.environ.get('STX_BENCH_DEBUG_ONE_GPU') == '1'
    if debug_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    from style_transfer_amd.netspec import builtin_net
    from style_transfer_amd.weights import synthetic_weights
    net = builtin_net('vgg19')
    rows, cols = GRIDS[world] if not opts.debug_grid else \
        tuple(int(v) for v in opts.debug_grid.split('x'))

    if world == 1:
        line = bench_single(opts, net, synthetic_weights(net, 0), local_rank, rows, cols)
    else:
        line = bench_ranks(opts, net, rank, world, local_rank, device, rows, cols, debug_one_gpu)
    if rank != 0:
        return
    devices = list(range(world)) if not debug_one_gpu else [0]
    if world > 1 and not opts.no_farm_leg and not debug_one_gpu:
        # north_star's layout on the same image and step loop: one host process, N GPUs.  The
        # other ranks have left (their process group is gone, their engines are closed).  A child
        # process with a deadline: a fault on a never-exercised peer path must not take the
        # benchmark line with it.
        line['farm'] = farm_leg_in_child(world, rows, cols, opts.steps, opts.warmup)
    if not opts.no_wall_clock and not debug_one_gpu:
        # the whole command-line run on this job's GPUs, one host process
        try:
            line.update(whole_run_wall_clock(devices))
        except Exception as err:      # pylint: disable=broad-except
            line['wall_clock_s'] = None
            line['wall_clock_error'] = '%s: %s' % (type(err).__name__, err)
    if world == 1 and not opts.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(net)
    print(json.dumps(line), flush=True)


def base_line(opts, world, rows, cols, elapsed, loss, eng, timed_group_ms):
    H, W = rows * TILE, cols * TILE
    tiles_per_step = rows * cols
    return {
        'metric': 'tile-iterations/sec, VGG-19 2048px/1024-tile (fwd+bwd, Gram/content losses, '
                  'regularizers, Adam step)',
        'value': tiles_per_step * opts.steps / elapsed,
        'unit': 'tile-iterations/s',
        'n_gpus': world, 'steps': opts.steps, 'warmup': opts.warmup,
        'ms_per_step': elapsed / opts.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'VGG-19 --size %d --tile-size %d -o adam: %dx%d image, %d tiles of '
                               '%dx%d per step, %d per GPU' % (max(H, W), TILE, W, H, tiles_per_step,
                                                             TILE, TILE, TILES_PER_GPU),
                   'content_layers': CONTENT_LAYERS, 'style_layers': STYLE_LAYERS,
                   'tiles_per_step': tiles_per_step, 'tiles_per_gpu': TILES_PER_GPU,
                   'final_loss': loss},
        'roofline': roofline_record(eng, float(np.mean(timed_group_ms))),
    }


def bench_single(opts, net, weights, device_index, rows, cols):
    """N = 1: the step loop through TileFarm (four engines = four HIP streams on the GPU)."""
    job = FarmJob(net, weights, [device_index], rows, cols)
    elapsed, loss = job.timed(opts.steps, opts.warmup)
    timed_group_ms = list(job.group_ms)
    line = base_line(opts, 1, rows, cols, elapsed, loss, job.eng, timed_group_ms)
    if opts.steady_seconds > 0:
        n_steady = max(1, int(np.ceil(opts.steady_seconds / (elapsed / opts.steps))))
        dt, _ = job.timed(n_steady, 0, clock_marks=True)
        add_clock(line['roofline'], job.clock)
        line['steady'] = {'steps': n_steady, 'seconds': dt, 'ms_per_step': dt / n_steady * 1e3,
                          'value': job.tiles_per_step * n_steady / dt, 'unit': 'tile-iterations/s'}
    line['graphs'] = job.graph_counters()
    job.close()
    return line


def bench_ranks(opts, net, rank, world, local_rank, device, rows, cols, debug_one_gpu):
    """N > 1: one process per GPU under torch.distributed.run.  Returns the line on rank 0."""
    import torch
    import torch.distributed as dist
    from style_transfer_amd import image_ops
    from style_transfer_amd.dist import DistributedTiles, broadcast_targets, broadcast_weights
    from style_transfer_amd.engine import DeviceArray, TileEngine
    from style_transfer_amd.farm import TileFarm, tile_grid
    from style_transfer_amd.optimizers import AdamOptimizer
    from style_transfer_amd.weights import synthetic_weights

    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo' if debug_one_gpu else 'nccl', rank=rank, world_size=world)
    # weights: built once on rank 0 and broadcast with ONE RCCL collective (80 MB); every other
    # rank sets its engines from the received device tensors
    weights = broadcast_weights(synthetic_weights(net, 0) if rank == 0 else None, device)
    # every rank: TILES_PER_GPU engines (HIP streams) on its GPU sharing one weight bank and one
    # target set, one tile of the step each
    engines = [TileEngine(net, local_rank, weights)]
    engines += [TileEngine(net, local_rank, share=engines[0]) for _ in range(TILES_PER_GPU - 1)]
    eng = engines[0]
    H, W = rows * TILE, cols * TILE
    rects = tile_grid((H, W), TILE)
    assert len(rects) % world == 0 and (opts.debug_grid or len(rects) == TILES_PER_GPU * world)
    tiles_per_rank = len(rects) // world
    content_weight = {'conv4_2': 0.05}
    style_weight = {l: 1.0 / len(STYLE_LAYERS) for l in STYLE_LAYERS}

    # ---- targets (once, outside the timed region): style Grams and the content map of the
    # image, computed on rank 0's GPU and broadcast device to device
    contents, styles = [], []
    if rank == 0:
        helper = TileFarm(net, verbose=False, engines=[eng])
        style_feats = helper.prepare_features_device(smooth_picture(7, TILE, TILE), STYLE_LAYERS,
                                                     TILE, passes=1)
        styles = [{l: eng.gram_matrix(f) for l, f in style_feats.items()}]
        contents = [helper.prepare_features_device(smooth_picture(8, H, W), CONTENT_LAYERS, TILE,
                                                   passes=1)]
    contents, styles = broadcast_targets(contents, styles, device)
    eng.set_contents_and_styles(contents, styles)          # (the rank's other engines share them)
    eng.sync()

    def wrap(tensor, engine):
        """A DeviceArray view of a torch tensor (no copy; torch keeps ownership)."""
        return DeviceArray.from_pointer(engine, tensor.data_ptr(), tensor.shape, owner=tensor)

    group_ms = []           # GPU time of one concurrent group of tile evaluations on this rank
    state = {}
    if rank == 0:
        rng = np.random.RandomState(0)
        img = eng.to_device(rng.uniform(0, 255, (3, H, W)).astype(np.float32) -
                            np.float32(MEAN).reshape(3, 1, 1))
        state.update(img=img, grad=eng.empty((3, H, W)),
                     old=eng.empty((3, H, W)).copy_from(img), rng=np.random.RandomState(0),
                     opt=AdamOptimizer(eng, img, step_size=15, bp1=1 - 1 / 20, decay=0.05,
                                       power=0.5))
    inflight = []
    grad_bufs = [torch.empty((3, TILE, TILE), dtype=torch.float32, device=device)
                 for _ in range(tiles_per_rank)]
    tile_bufs = {}

    def evaluate_begin(jobs, roll):
        """Enqueues this rank's tiles, one per engine (they run concurrently)."""
        inflight.clear()
        for k, (tile, start) in enumerate(jobs):
            e = engines[k % len(engines)]
            inflight.append((e, k, e.sc_grad_tile_async(
                wrap(tile, e), start, roll, CONTENT_LAYERS, STYLE_LAYERS, {}, content_weight,
                style_weight, grad_out=wrap(grad_bufs[k], e))))

    def evaluate_end():
        """Waits for them; [(loss, grad tensor)]."""
        used = []
        for e, _, _ in inflight:
            if e not in used:
                used.append(e)
        for e in used:
            e.sync()
        group_ms.append(max(e.last_tile_ms() for e in used))
        return [(p.loss, grad_bufs[k]) for _, k, p in inflight]

    def cut(rect, roll):
        if rect not in tile_bufs:
            tile_bufs[rect] = torch.empty((3, rect[1] - rect[0], rect[3] - rect[2]),
                                          dtype=torch.float32, device=device)
        image_ops.cut_tile(eng, state['img'], roll, rect, wrap(tile_bufs[rect], eng))
        return tile_bufs[rect]

    def put(rect, g, roll):
        image_ops.put_tile(eng, state['grad'], roll, rect, wrap(g, eng))

    farm = DistributedTiles(lambda rect, roll: (cut(rect, roll), eng.sync())[0],
                            (evaluate_begin, evaluate_end), put, device)

    def step():
        if rank != 0:
            farm.eval_sc_grad(rects, (0, 0))
            return None
        xy = np.int32(state['rng'].uniform(-0.5, 0.5, size=2) * (H, W)) // 8
        roll = xy * 8

        def opfunc(params):
            loss = farm.eval_sc_grad(rects, roll)
            reg = image_ops.regularizers(eng, params, state['grad'], MEAN, 5.0, 2.0, 2.0, 6.0)
            eng.sync()
            return loss + reg.value, state['grad']
        avg, loss = state['opt'].update(opfunc)
        image_ops.step_stats(eng, avg, state['old'])
        return loss

    def fence():
        for e in engines:
            e.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    wire = 'cpu' if debug_one_gpu else device
    loss = None
    for _ in range(opts.warmup):
        step()
    fence()
    group_ms.clear()
    t0 = time.perf_counter()
    for _ in range(opts.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=wire)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    timed_group_ms = list(group_ms)

    # ---- a second, longer measurement of the same loop (the timed region above is short)
    steady = None
    clock = None
    if opts.steady_seconds > 0:
        t = torch.tensor([max(1, int(np.ceil(opts.steady_seconds / (elapsed / opts.steps))))],
                         dtype=torch.int64, device=wire)
        dist.broadcast(t, 0)
        n_steady = int(t[0])
        fence()
        if rank == 0:
            for e in engines:
                e.clock_marks(True)
        t1 = time.perf_counter()
        for _ in range(n_steady):
            step()
        fence()
        dt = time.perf_counter() - t1
        if rank == 0:
            clock = clock_summary(engines)
        steady = {'steps': n_steady, 'seconds': dt, 'ms_per_step': dt / n_steady * 1e3,
                  'value': len(rects) * n_steady / dt, 'unit': 'tile-iterations/s'}
    line = None
    if rank == 0:
        line = base_line(opts, world, rows, cols, elapsed, loss, eng, timed_group_ms)
        add_clock(line['roofline'], clock)
        if steady is not None:
            line['steady'] = steady
    # every rank lets go of its GPU before rank 0 goes on alone (farm and whole-run legs drive
    # all GPUs from one process); no collective is pending past this point
    dist.barrier()
    dist.destroy_process_group()
    for e in engines:
        e.close()
    return line


if __name__ == '__main__':
    main()
