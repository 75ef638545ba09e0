#!/usr/bin/env python3
"""Headline benchmark: tile-iterations/s of the tiled style-transfer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one optimizer iteration of the reference's step loop (style_transfer.py:771-806) at the
top scale of the configuration BASELINE.json's metric is quoted on ("VGG-19 --size 2048 --tile-size
1024, Adam"): draw the seam-suppression shift, cut the four 1024 x 1024 tiles, per-tile VGG-19
forward + Gram/content losses + backward (stx_sc_grad_tile), stitch, TV + p-norm regularizers,
fused Adam step with iterate averaging, step statistics.  Everything is resident in HBM when the
timed region starts.  Synthetic data: seeded He-initialised VGG-19 weights, seeded low-pass-noise
content and style pictures (no network).

`value` is ALWAYS that workload -- the fixed 2048 x 2048 image, four tiles per step -- so for
N > 1 it is STRONG scaling, exactly what the metric names ("VGG-19 2048px/1024-tile, 1/2/4/8
MI355X"): 2 / 1 / 1 tiles per GPU at N = 2 / 4 / 8 (at N = 8 four GPUs have no tile and the line
says so).  At N = 1 the step loop runs through TileFarm, the product's own driver (two HIP
streams on the GPU, TileFarm's default; STX_STREAMS_PER_GPU overrides it for A/B runs).  At N > 1 it runs one process per GPU as the benchmark contract prescribes:
rank 0 owns the image and the optimizer, tiles go out and gradients come back as batched
point-to-point transfers over RCCL, no collective on the data path.

Sub-records at N > 1 (each also carries `bit_identical`: step 1 of its warm-up is evaluated on the
N GPUs and again on GPU 0 alone -- tiles are independent, so loss and gradient must agree bit for
bit; a run on real xGMI thereby validates the peer copies / RCCL transfers by itself):
  farm          north_star's layout on the same 2048 x 2048 workload: ONE host process driving
                the N GPUs through TileFarm (tiles and gradients as xGMI peer copies ordered by
                events, no host wait inside a step);
  config4       BASELINE config 4's top scale: 4096 x 4096, 16 tiles of 1024 x 1024, L-BFGS,
                through TileFarm over the N GPUs;
  weak          the round-1..3 headline kept for continuity: four tiles per GPU, the image grows
                with N (2048 x 4096, 4096 x 4096, 4096 x 8192), one process per GPU.

The line also carries
  roofline      the time the kernels' own instructions need on the matrix pipe, in fp32-MFMA FLOP,
                for the concurrent tile-iterations of one GPU (fp32 convolutions through Winograd
                F(2x2,3x3) issue 4/9 of a direct convolution's MFMAs; the fp16-split convolutions,
                1-D Winograd F(2,3) with three fp16 MFMAs of 1/16 of an fp32 MFMA's time per k,
                6/9 x 3/16 = 1/8 of it; Gram and SYMM run as six bf16 MFMAs per 16 k and are
                counted at the 0.375 of their fp32 pipe time that this occupies) over the GPU time of that concurrent group of stx_sc_grad_tile calls --
                HIP events on each engine's own stream inside the timed region, the longest of the
                spans -- against the fp32 MFMA peak, so that frac <= 1 by construction;
                `frac_driver_clock` divides the same work by the wall-clock ms_per_step instead;
                `bound_ms` is the time the same work takes at that peak.  The SURVEY 8d figure
                (1 514 240 FLOP per tile pixel, every convolution counted as a direct one) is kept
                beside it as achieved_direct_equiv;
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
PEAK_FP16_MFMA_TFLOPS = 16 * PEAK_FP32_MFMA_TFLOPS    # v_mfma_f32_32x32x16_f16: 16 k per 32 cycles against 2 k per 64 (the guide's ~2.5 PFLOP/s dense)
NOMINAL_CLOCK_MHZ = 2400.0
# the arithmetic the path computes in: float32 values throughout; the 3x3 layers from 128 channels up
# form their products on the fp16 matrix cores from two-piece operands (22 significand bits) and add
# them in fp32 -- every parity bound of tests/ is the float32 kernels' own (STX_CONV_H2=0: fp32 MFMAs only)
DTYPE = ('f32' if os.environ.get('STX_CONV_H2') == '0' and not os.environ.get('STX_CONV_H2_BWD')
         else 'f32 (fp16x2-split MFMA, fp32 accumulate)')
STREAMS_PER_GPU = int(os.environ.get('STX_STREAMS_PER_GPU', '2'))    # engines (HIP streams) a GPU runs its tiles of a step on
STRONG_GRID = (2, 2)                   # BASELINE's metric: --size 2048 --tile-size 1024
CONFIG4_GRID = (4, 4)                  # BASELINE config 4's top scale: --size 4096, 16 tiles
WEAK_GRIDS = {1: (2, 2), 2: (2, 4), 4: (4, 4), 8: (4, 8)}    # four tiles per GPU
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


TRAFFIC_PROFILE = 'profiles/r06_hbm_traffic_pmc.json'
TRAFFIC_PROFILE_FALLBACK = 'profiles/r05b_hbm_traffic_pmc.json'


def measured_traffic(tiles_per_launch):
    """(bytes per launch group, source file): HBM bytes from the PMC passes over THIS script
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per
    the gfx950 calibration on the Adam kernel -- tools/pmc_traffic.py).  Counters cannot be
    collected inside a timed run, so the figure is a replay of the committed profile of this
    build state and the line names the file; (None, None) if it is missing."""
    for name in (TRAFFIC_PROFILE, TRAFFIC_PROFILE_FALLBACK):
        try:
            with open(os.path.join(REPO, name)) as f:
                # measured per tile-iteration
                return float(json.load(f)['hbm_bytes_per_tile_iteration']) * tiles_per_launch, name
        except (OSError, KeyError, ValueError):
            continue
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


CONTENT_WEIGHT = {'conv4_2': 0.05}
STYLE_WEIGHT = {l: 1.0 / len(STYLE_LAYERS) for l in STYLE_LAYERS}


def start_image(H, W):
    """The reference's start image: uniform noise minus the mean (style_transfer.py:889)."""
    rng = np.random.RandomState(0)
    return rng.uniform(0, 255, (3, H, W)).astype(np.float32) - np.float32(MEAN).reshape(3, 1, 1)


def draw_roll(rng, H, W):
    """The iteration's seam-suppression shift (style_transfer.py:777-779)."""
    return (np.int32(rng.uniform(-0.5, 0.5, size=2) * (H, W)) // 8) * 8


def targets_on(farm, H, W):
    """Style Grams of a 1024 x 1024 picture and the content map of the H x W picture, computed on
    the farm's master GPU."""
    style_feats = farm.prepare_features_device(smooth_picture(7, TILE, TILE), STYLE_LAYERS, TILE,
                                               passes=1)
    styles = [{l: farm.master.gram_matrix(f) for l, f in style_feats.items()}]
    contents = [farm.prepare_features_device(smooth_picture(8, H, W), CONTENT_LAYERS, TILE, passes=1)]
    return contents, styles


class FarmJob:
    """The benchmark's step loop on a TileFarm over `devices` (one host process): image of
    rows x cols tiles of 1024 x 1024, targets computed on the master GPU, Adam (or L-BFGS)."""

    def __init__(self, net, weights, devices, rows, cols, optimizer='adam', force_staging=None):
        from style_transfer_amd import image_ops
        from style_transfer_amd.farm import TileFarm
        from style_transfer_amd.optimizers import AdamOptimizer, LBFGSOptimizer
        self.image_ops = image_ops
        self.net, self.weights = net, weights
        self.H, self.W = rows * TILE, cols * TILE
        self.farm = TileFarm(net, list(devices), weights, verbose=False,
                             streams_per_device=STREAMS_PER_GPU, force_staging=force_staging)
        eng = self.eng = self.farm.master
        self.farm.set_contents_and_styles(*targets_on(self.farm, self.H, self.W))
        self.img = eng.to_device(start_image(self.H, self.W))
        self.grad = eng.empty((3, self.H, self.W))
        self.old = eng.empty((3, self.H, self.W)).copy_from(self.img)
        self.rng = np.random.RandomState(0)
        if optimizer == 'adam':
            self.opt = AdamOptimizer(eng, self.img, step_size=15, bp1=1 - 1 / 20, decay=0.05, power=0.5)
        else:       # the command line's -o lbfgs (style_transfer.py:896-897)
            self.opt = LBFGSOptimizer(eng, self.img)
        self.group_ms = []
        self.clock = None
        self.tiles_per_step = rows * cols
        self.tiles_per_gpu = -(-self.tiles_per_step // len(set(devices)))
        # Adam: the host queues iteration i + 1 before it collects the loss and the statistics of
        # iteration i (StyleTransfer.transfer does the same: a fence per iteration); L-BFGS has a
        # host decision inside its step and stays synchronous
        self.run_ahead = optimizer == 'adam' and os.environ.get('STX_RUN_AHEAD', '1') != '0'
        self.in_flight = None
        self.last_loss = None

    def _finish(self, item):
        loss, stats = item
        self.last_loss = float(loss)          # waits for that iteration's fences only
        stats.values()
        self.group_ms.append(group_span_ms(self.farm.engines, self.tiles_per_step))

    def drain(self):
        if self.in_flight is not None:
            self._finish(self.in_flight)
            self.in_flight = None
        return self.last_loss

    def step(self):
        """One iteration of the reference's step loop (style_transfer.py:771-815).  Returns the
        loss of the newest FINISHED iteration (run-ahead: the previous one; drain() the last)."""
        roll = draw_roll(self.rng, self.H, self.W)

        def opfunc(params):
            loss = self.farm.eval_sc_grad(params, self.grad, roll, CONTENT_LAYERS, STYLE_LAYERS, {},
                                          CONTENT_WEIGHT, STYLE_WEIGHT, TILE, lazy=True)
            loss.add(self.image_ops.regularizers(self.eng, params, self.grad, MEAN, 5.0, 2.0, 2.0,
                                                 6.0), self.eng)
            return loss, self.grad
        avg, loss = self.opt.update(opfunc)
        if self.run_ahead:
            stats = self.image_ops.step_stats_async(self.eng, avg, self.old)
            loss.seal(also=[self.eng])
            previous, self.in_flight = self.in_flight, (loss, stats)
            if previous is not None:
                self._finish(previous)
            return self.last_loss
        self.image_ops.step_stats(self.eng, avg, self.old)
        self.last_loss = float(loss)
        self.group_ms.append(group_span_ms(self.farm.engines, self.tiles_per_step))
        return self.last_loss

    def fence(self):
        self.drain()
        for e in self.farm.engines:
            e.sync()

    def timed(self, steps, warmup, clock_marks=False):
        """W untimed iterations, everything finished; then exactly K iterations, all of them
        finished (losses collected, every engine synchronised) when the clock stops."""
        for _ in range(warmup):
            self.step()
        self.fence()
        self.group_ms.clear()
        # (clock_marks: one workgroup of every 2-D Winograd launch times its chunk loop with the core
        # and the 100 MHz counters -- the second, longer measurement only, never the headline's steps)
        engines = self.farm.engines[:self.tiles_per_step] if clock_marks else []
        for e in engines:
            e.clock_marks(True)
        evals0 = self.farm.tile_evals
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        loss = self.last_loss
        elapsed = time.perf_counter() - t0
        self.timed_tile_evals = self.farm.tile_evals - evals0
        if clock_marks:
            self.clock = clock_summary(engines)
        return elapsed, loss

    def bit_identical(self):
        """Evaluates ONE step's loss and gradient on this farm (all its GPUs) and on a one-GPU farm
        on the master's device: tiles are independent, so both must be bit-identical.  Over real
        xGMI this validates the peer copies and the event ordering by itself."""
        from style_transfer_amd.farm import TileFarm
        roll = draw_roll(np.random.RandomState(123), self.H, self.W)
        args = (roll, CONTENT_LAYERS, STYLE_LAYERS, {}, CONTENT_WEIGHT, STYLE_WEIGHT, TILE)
        loss_n = self.farm.eval_sc_grad(self.img, self.grad, *args)
        grad_n = self.grad.get()
        solo = TileFarm(self.net, [self.eng.device], verbose=False, engines=[self.eng])
        ref = self.eng.empty((3, self.H, self.W))
        loss_1 = solo.eval_sc_grad(self.img, ref, *args)
        same = bool(loss_n == loss_1 and np.array_equal(grad_n, ref.get()))
        ref.free()
        solo.close()
        return same

    def close(self):
        self.farm.close()


def farm_leg(devices, rows, cols, steps, warmup, optimizer='adam', force_staging=None, check=True):
    """The benchmark's step loop through TileFarm over `devices` in THIS process: the `farm` and
    `config4` sub-records."""
    from style_transfer_amd import lib
    from style_transfer_amd.netspec import builtin_net
    from style_transfer_amd.weights import synthetic_weights
    net = builtin_net('vgg19')
    job = FarmJob(net, synthetic_weights(net, 0), devices, rows, cols, optimizer, force_staging)
    same = job.bit_identical() if check else None
    elapsed, loss = job.timed(steps, warmup)
    n_dev = len(set(devices))
    busy = min(n_dev, job.tiles_per_step)
    record = {'layout': 'one host process, TileFarm over %d GPU(s) (xGMI peer copies, event-ordered, '
                        'no host wait inside a step)' % n_dev,
              'workload': 'VGG-19 %dx%d image, %d tiles of %dx%d per step, -o %s; %d tile(s) per busy GPU, '
                          '%d GPU(s) without a tile' % (job.W, job.H, job.tiles_per_step, TILE, TILE,
                                                         optimizer, job.tiles_per_gpu, n_dev - busy),
              'scaling': 'strong', 'n_gpus': n_dev, 'idle_gpus': n_dev - busy,
              'value': job.timed_tile_evals / elapsed, 'unit': 'tile-iterations/s',
              'ms_per_step': elapsed / steps * 1e3, 'steps': steps, 'final_loss': loss,
              'avg_launch_ms': float(np.mean(job.group_ms)),
              'tile_evals': int(sum(e.query(lib.Q_TILE_EVALS) for e in job.farm.engines)),
              'peers_without_access': int(sum(e.query(lib.Q_PEERS_WITHOUT_ACCESS)
                                              for e in job.farm.primaries())),
              'bit_identical': same}
    job.close()
    return record


def farm_leg_in_child(devices, rows, cols, steps, warmup, optimizer='adam', force_staging=False,
                      timeout=420):
    """farm_leg in a child process with a deadline: a fault on a never-exercised peer path must
    not take the benchmark line with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--farm-leg', ','.join(str(d) for d in devices),
           '--debug-grid', '%dx%d' % (rows, cols), '--steps', str(steps), '--warmup', str(warmup),
           '--farm-optimizer', optimizer]
    if force_staging:
        cmd.append('--farm-force-staging')
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
F16X2_PIPE_TIME = 3 * 32 / (8 * 64)     # three fp16 MFMAs per 16 k (two-piece operands, round 5)


def terms_pipe_time():
    """Pipe time of a tile's Gram + SYMM products as a fraction of their fp32-MFMA form's: the two-piece
    fp16 kernels by default (STX_GRAM / STX_SYMM = bf3 | fp32: the three-piece bf16 / the fp32 kernels);
    the Gram of conv1_1 is computed by the first layer's kernel with three bf16 pieces (4 of the 17 quarter
    units of the five Gram products: 4 + 4 + 4 + 4 + 1)."""
    rate = {'fp32': 1.0, 'bf3': BF16X3_PIPE_TIME}
    gram = rate.get(os.environ.get('STX_GRAM'), F16X2_PIPE_TIME)
    symm = rate.get(os.environ.get('STX_SYMM'), F16X2_PIPE_TIME)
    first = BF16X3_PIPE_TIME if os.environ.get('STX_CONV_FIRST_FUSED') != '0' else gram
    return 0.5 * (first * 4 / 17 + gram * 13 / 17) + 0.5 * symm


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


def group_span_ms(engines, tiles_per_step):
    """GPU time of one step's concurrent group of tile evaluations: tile t runs on engine t mod n, the
    tiles of one engine one after the other on its stream -- an engine with k tiles of the step
    counts k x the HIP-event span of its newest call (stx_last_tile_ms times one call; the calls of a
    step are alike); the longest engine is the group."""
    n = min(len(engines), tiles_per_step)
    return max(engines[i].last_tile_ms() * len(range(i, tiles_per_step, n)) for i in range(n))


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


def roofline_record(eng, avg_group_ms, tiles_per_gpu, ms_per_step):
    """Matrix-pipe work of one GPU's concurrent tile evaluations over their HIP-event span,
    against the fp32 MFMA peak.  Every term is the time the kernels' own instructions need on the
    matrix pipe, expressed in fp32-MFMA FLOP: fp32 Winograd convolutions issue 4/9 (2-D) or 2/3 (1-D)
    of a direct convolution's fp32 MFMAs; the fp16-split convolutions (conv_h2.hip) three fp16 MFMAs
    per 16 k for 6 of every 9 multiplies, i.e. 1/8 of a direct convolution's fp32 pipe time; Gram and
    SYMM issue three fp16 MFMAs per 16 k (round 5; the Gram of conv1_1 six bf16 ones), which occupy the
    pipe for 0.1875 (0.375) of the time their fp32 form would -- so frac <= 1 by construction."""
    flop = FLOP_PER_TILE_PIXEL * TILE * TILE * tiles_per_gpu
    direct_equiv = flop / (avg_group_ms * 1e-3) / 1e12
    conv_alg, conv_issued = eng.last_tile_flops()
    terms = GRAM_SYMM_FLOP_PER_TILE_PIXEL * TILE * TILE
    per_tile = flop / tiles_per_gpu - conv_alg - terms + conv_issued
    issued_r02 = (per_tile + terms) * tiles_per_gpu            # round-2 accounting: terms at the fp32 rate
    issued = (per_tile + terms * terms_pipe_time()) * tiles_per_gpu
    issued_tflops = issued / (avg_group_ms * 1e-3) / 1e12
    bound_ms = issued / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
    traffic, traffic_src = measured_traffic(tiles_per_gpu)
    return {'bound': 'mfma', 'achieved': issued_tflops, 'peak': PEAK_FP32_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': issued_tflops / PEAK_FP32_MFMA_TFLOPS,
            # the same fraction stated against the pipe it is really measured on: pipe time is pipe time, so
            # x fp32-MFMA-equivalent TFLOP/s of 157.3 = 16 x of the fp16 pipe's 2 516.8 (with the fp16-split
            # kernels nearly all of the issued work IS fp16 MFMAs)
            'peak_fp16': PEAK_FP16_MFMA_TFLOPS, 'achieved_fp16_equiv': 16 * issued_tflops,
            'frac_fp16_pipe': issued_tflops / PEAK_FP32_MFMA_TFLOPS,
            # the same work over the wall-clock step (cut, stitch, regularizers, Adam, statistics
            # and the host's share included): what the driver's own clock sees
            'frac_driver_clock': issued / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'frac_round2_accounting': issued_r02 / (avg_group_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'bound_ms': bound_ms, 'bound_ms_per_tile': bound_ms / tiles_per_gpu,
            'traffic': traffic, 'traffic_unit': 'bytes per launch',
            'traffic_source': traffic_src,
            'kernel': 'stx_sc_grad_tile x %d concurrent on one GPU: conv_h2_kernel<0|1|3, 1|2, 1|2> (every 3x3 '
                      'layer from 64 input channels, forward / backward / loss-injecting backward: fp16 MFMA, '
                      'two-piece operand split, fp32 accumulate; three quarters of the kernel time), '
                      'conv_first_kernel (first layer + its Gram partials), conv3x3_m4_kernel (backward into '
                      'the image), gram_partial_h2_kernel / symm_h2_kernel (fp16 MFMA, two-piece split)'
                      % tiles_per_gpu,
            'flop_issued_per_launch': issued, 'avg_launch_ms': avg_group_ms,
            'achieved_direct_equiv': direct_equiv,
            'flop_direct_equiv_per_launch': flop,
            'note': 'achieved / frac = the time the issued matrix instructions need on the matrix pipe, '
                    'priced in fp32-MFMA FLOP (fp32 Winograd F(2x2,3x3) convolutions issue 4/9 of a direct '
                    'convolution; the fp16-split 1-D Winograd convolutions three fp16 MFMAs of 1/16 of an '
                    'fp32 MFMA\'s time per k for 6 of 9 multiplies = 1/8 of a direct convolution\'s pipe '
                    'time; Gram and SYMM three fp16 MFMAs per 16 k = 0.1875 of the pipe time of their fp32 '
                    'form, the Gram of conv1_1 six bf16 ones = 0.375) over the HIP-event time of the launch '
                    'group, against the fp32 '
                    'MFMA peak at 2.4 GHz: the fraction of that time the matrix pipes are busy at the '
                    'nominal clock; frac_driver_clock = the same work over the wall-clock '
                    'ms_per_step; clock_mhz is the shader clock INSIDE the Winograd convolution '
                    'kernels (fp32 and fp16-split) during the second, longer measurement (`steady`; stx_clock_marks: one '
                    'workgroup of every launch reads core cycles and the 100 MHz counter around its '
                    'chunk loop; the median over the launches, with the 10th / 90th percentile), '
                    'peak_at_clock the fp32 MFMA peak at min(that clock, 2.4 GHz) and frac_at_clock = '
                    'achieved / peak_at_clock: the part is power-limited under matrix load -- 1.4-1.7 GHz '
                    'inside the fp16-split kernels, 1.9-2.2 GHz inside the fp32 ones -- and does not hold '
                    'its 2.4 GHz there; '
                    'frac_round2_accounting counts Gram and SYMM in full as '
                    'round 2 did (they ran on the fp32 pipe then); achieved_direct_equiv credits '
                    'every convolution as a direct one (SURVEY 8d: 1 514 240 FLOP per tile pixel) '
                    'and is not a roofline fraction; traffic is a replay of the committed PMC '
                    'profile named in traffic_source, not a measurement of this run'}


def kernel_records(job, reps=5):
    """Per-kernel-group figures of ONE 1024 x 1024 tile evaluation on one stream, measured in this run
    (stx_profile_read: HIP events around every launch group inside the engine), after the timed legs:
    `dominant` -- the kernel that takes most of a tile's time, conv_h2_kernel<0,2,1,0> (the forward 3x3 layers
    from 128 input channels: conv2_2 .. conv5_1) --, `furthest_below` -- the launch group with the lowest
    fraction of its pipe --, and the groups' shares.  Issued FLOP of an fp16-split layer = 2 x its direct
    count (6 of 9 multiplies x 3 fp16 products), against the fp16 MFMA peak."""
    eng = job.eng
    tile = eng.to_device(smooth_picture(9, TILE, TILE))
    grad = eng.empty((3, TILE, TILE))
    args = ((0, 0), (0, 0), CONTENT_LAYERS, STYLE_LAYERS, {}, CONTENT_WEIGHT, STYLE_WEIGHT)
    for _ in range(2):
        eng.sc_grad_tile_async(tile, *args, grad_out=grad)
    eng.sync()
    eng.profile(True)
    acc, order = {}, []
    for _ in range(reps):
        eng.sc_grad_tile_async(tile, *args, grad_out=grad)
        for label, ms, flops in eng.profile_read():
            if label not in acc:
                acc[label] = [0.0, flops]
                order.append(label)
            acc[label][0] += ms / reps
    eng.profile(False)
    tile.free()
    grad.free()
    h2_min = int(os.environ.get('STX_CONV_H2', '64') or 0)
    groups = {}

    def add(group, label, ms, issued):
        g = groups.setdefault(group, {'launch_groups': [], 'ms_per_tile': 0.0, 'flop_issued_fp16': 0.0})
        g['launch_groups'].append(label)
        g['ms_per_tile'] += ms
        g['flop_issued_fp16'] += issued
    total = 0.0
    for label in order:
        ms, direct = acc[label]
        total += ms
        kind, _, layer = label.partition(' ')
        if kind in ('fwd', 'bwd') and layer.startswith('conv') and direct > 0:
            scale, cout = eng.layer_info(layer)
            cin = int(round(direct / (18.0 * cout * (TILE // scale) ** 2)))
            k_in = cin if kind == 'fwd' else cout          # channels the launch reduces over
            if min(cin, cout) < 8:
                add('first layer / backward into the image (fp32 MFMA)', label, ms, 0.0)
            elif h2_min and k_in >= max(h2_min, 128) and min(cin, cout) >= 128:
                # (a backward launch adds the loss terms of the blob whose gradient it writes: convX_Y's is convX_(Y-1))
                x, _, y = layer[4:].partition('_')
                inject = kind == 'bwd' and 'conv%s_%d' % (x, int(y) - 1) in STYLE_LAYERS + CONTENT_LAYERS
                add('%s 3x3 layers from 128 channels (conv_h2_kernel<%s,2,1,*>)'
                    % ('forward' if kind == 'fwd' else 'loss-injecting backward' if inject else 'backward',
                       '0' if kind == 'fwd' else '3' if inject else '1'), label, ms, 2 * direct)
            elif h2_min and k_in >= h2_min:
                add('64-channel layers: %s (conv_h2_kernel, 64-channel or two-patch tilings)' % label, label, ms, 2 * direct)
            else:
                add('fp32 convolution: ' + label, label, ms, 0.0)
        else:
            add('loss terms, pooling, copies', label, ms, 0.0)
    rows = []
    for name, g in groups.items():
        tf = g['flop_issued_fp16'] / (g['ms_per_tile'] * 1e-3) / 1e12 if g['ms_per_tile'] > 0 else 0.0
        rows.append({'name': name, 'launch_groups': len(g['launch_groups']), 'ms_per_tile': g['ms_per_tile'],
                     'share_of_tile': g['ms_per_tile'] / total, 'flop_issued_fp16': g['flop_issued_fp16'],
                     'tflops_issued': tf, 'frac_fp16_pipe': tf / PEAK_FP16_MFMA_TFLOPS})
    matrix = [r for r in rows if r['flop_issued_fp16'] > 0]
    out = {'tile_ms_single_stream': total, 'groups': rows,
           'note': 'one tile evaluation alone on one stream (the timed steps run two at a time): HIP events around '
                   'every launch group, mean of %d evaluations; issued FLOP of an fp16-split layer = 2 x direct; '
                   'frac_fp16_pipe against %.1f TFLOP/s' % (reps, PEAK_FP16_MFMA_TFLOPS)}
    if matrix:
        out['dominant'] = max(matrix, key=lambda r: r['ms_per_tile'])
        out['furthest_below'] = min(matrix, key=lambda r: r['frac_fp16_pipe'])
    return out


FP32_KERNELS = {'STX_CONV_H2': '0', 'STX_GRAM': 'fp32', 'STX_SYMM': 'fp32', 'STX_STREAMS_PER_GPU': '4'}


def fp32_kernels_leg(opts, net, weights, device_index, rows, cols):
    """The same step loop with the fp32-MFMA kernels only (no fp16-split convolution, Gram or SYMM: round 4's
    arithmetic, on the four streams per GPU that were its best) -- the strict-fp32 figure, timed in the same
    run on the same box (stx_reread_env: the library takes a new snapshot of its switches)."""
    global STREAMS_PER_GPU
    old_env = {k: os.environ.get(k) for k in FP32_KERNELS}
    old_streams = STREAMS_PER_GPU
    from style_transfer_amd import lib
    os.environ.update(FP32_KERNELS)
    lib.reread_env()
    STREAMS_PER_GPU = 4
    try:
        job = FarmJob(net, weights, [device_index], rows, cols)
        elapsed, loss = job.timed(opts.steps, opts.warmup)
        group_ms = float(np.mean(job.group_ms))
        conv_alg, conv_issued = job.eng.last_tile_flops()
        job.close()
    finally:
        STREAMS_PER_GPU = old_streams
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.reread_env()
    return {'value': rows * cols * opts.steps / elapsed, 'unit': 'tile-iterations/s',
            'ms_per_step': elapsed / opts.steps * 1e3, 'steps': opts.steps, 'warmup': opts.warmup,
            'dtype': 'f32', 'switches': FP32_KERNELS, 'final_loss': loss, 'avg_launch_ms': group_ms,
            'conv_flop_issued_over_direct': conv_issued / conv_alg if conv_alg else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-wall-clock', action='store_true',
                    help='skip the whole-run wall-clock leg (about 10 s)')
    ap.add_argument('--no-farm-leg', action='store_true',
                    help='N > 1: skip the single-host-process (TileFarm) sub-records `farm` and `config4`')
    ap.add_argument('--no-weak', action='store_true', help='N > 1: skip the weak-scaling sub-record')
    ap.add_argument('--farm-leg', default='', metavar='D0,D1,...',
                    help='internal: only the single-host-process (TileFarm) step loop over these devices '
                         'with the tile grid of --debug-grid; prints the sub-record')
    ap.add_argument('--farm-optimizer', default='adam', choices=['adam', 'lbfgs'])
    ap.add_argument('--farm-force-staging', action='store_true')
    ap.add_argument('--steady-seconds', type=float, default=5.0)
    ap.add_argument('--no-kernel-records', action='store_true',
                    help='N = 1: skip roofline.per_kernel (five profiled tile evaluations after the timed legs)')
    ap.add_argument('--no-fp32-leg', action='store_true',
                    help='N = 1: skip the fp32_kernels sub-record (the same steps with the fp32-MFMA kernels only)')
    ap.add_argument('--debug-grid', default=None,
                    help='RxC tile grid instead of the 2x2 of the metric (tests only: lets a single '
                         'process evaluate the image of a larger job)')
    opts = ap.parse_args()

    if opts.farm_leg:
        r, c = (int(v) for v in opts.debug_grid.split('x'))
        devices = [int(d) for d in opts.farm_leg.split(',')]
        print(json.dumps(farm_leg(devices, r, c, opts.steps, opts.warmup, opts.farm_optimizer,
                                  opts.farm_force_staging or None)), flush=True)
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
    if opts.gpus not in WEAK_GRIDS:
        sys.exit('--gpus must be one of %s' % sorted(WEAK_GRIDS))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an AMD GPU (no CPU path exists)')
    # STX_BENCH_DEBUG_ONE_GPU=1 (debugging the N > 1 protocol on a 1-GPU box only): every rank
    # uses GPU 0, tiles travel over gloo through host memory, and the TileFarm sub-records list
    # GPU 0 N times with the cross-GPU staging leg forced.  Never a benchmark number.
    debug_one_gpu = os.environ.get('STX_BENCH_DEBUG_ONE_GPU') == '1'
    if debug_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    from style_transfer_amd.netspec import builtin_net
    from style_transfer_amd.weights import synthetic_weights
    net = builtin_net('vgg19')
    rows, cols = STRONG_GRID if not opts.debug_grid else \
        tuple(int(v) for v in opts.debug_grid.split('x'))

    if world == 1:
        line = bench_single(opts, net, synthetic_weights(net, 0), local_rank, rows, cols)
    else:
        line = bench_ranks(opts, net, rank, world, local_rank, device, rows, cols, debug_one_gpu)
    if rank != 0:
        return
    devices = list(range(world)) if not debug_one_gpu else [0] * world
    if world > 1 and not opts.no_farm_leg:
        # north_star's layout: one host process, N GPUs.  The other ranks have left (their process
        # group is gone, their engines are closed).
        line['farm'] = farm_leg_in_child(devices, rows, cols, opts.steps, opts.warmup,
                                         force_staging=debug_one_gpu)
        line['config4'] = farm_leg_in_child(devices, CONFIG4_GRID[0], CONFIG4_GRID[1],
                                            max(2, opts.steps // 2), min(opts.warmup, 2), 'lbfgs',
                                            force_staging=debug_one_gpu)
    if not opts.no_wall_clock:
        # the whole command-line run on this job's GPUs, one host process
        try:
            line.update(whole_run_wall_clock(devices))
        except Exception as err:      # pylint: disable=broad-except
            line['wall_clock_s'] = None
            line['wall_clock_error'] = '%s: %s' % (type(err).__name__, err)
    if world == 1 and not opts.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(net)
    print(json.dumps(line), flush=True)


def base_line(opts, world, rows, cols, elapsed, loss, eng, timed_group_ms, scaling='strong'):
    H, W = rows * TILE, cols * TILE
    tiles_per_step = rows * cols
    busy = min(world, tiles_per_step)
    tiles_per_gpu = -(-tiles_per_step // world)
    ms_per_step = elapsed / opts.steps * 1e3
    return {
        'metric': 'tile-iterations/sec, VGG-19 2048px/1024-tile (fwd+bwd, Gram/content losses, '
                  'regularizers, Adam step)',
        'value': tiles_per_step * opts.steps / elapsed,
        'unit': 'tile-iterations/s',
        'n_gpus': world, 'steps': opts.steps, 'warmup': opts.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': scaling,
        'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
        'config': {'workload': 'VGG-19 --size %d --tile-size %d -o adam: %dx%d image, %d tiles of '
                               '%dx%d per step, %d per busy GPU%s'
                               % (max(H, W), TILE, W, H, tiles_per_step, TILE, TILE, tiles_per_gpu,
                                  '' if busy == world else ' (%d of the %d GPUs have no tile: the '
                                  'workload has only %d)' % (world - busy, world, tiles_per_step)),
                   'content_layers': CONTENT_LAYERS, 'style_layers': STYLE_LAYERS,
                   'tiles_per_step': tiles_per_step, 'tiles_per_gpu': tiles_per_gpu,
                   'idle_gpus': world - busy, 'final_loss': loss},
        'roofline': roofline_record(eng, float(np.mean(timed_group_ms)), tiles_per_gpu, ms_per_step),
    }


def bench_single(opts, net, weights, device_index, rows, cols):
    """N = 1: the step loop through TileFarm (STREAMS_PER_GPU engines = HIP streams on the GPU)."""
    from style_transfer_amd import lib
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
    line['tile_evals'] = int(sum(e.query(lib.Q_TILE_EVALS) for e in job.farm.engines))
    if not opts.no_kernel_records:
        line['roofline']['per_kernel'] = kernel_records(job)
        if 'dominant' in line['roofline']['per_kernel']:
            line['roofline']['dominant'] = line['roofline']['per_kernel']['dominant']
    job.close()
    if not opts.no_fp32_leg and DTYPE != 'f32':
        line['fp32_kernels'] = fp32_kernels_leg(opts, net, weights, device_index, rows, cols)
    return line


class RankJob:
    """The step loop of one image with one process per GPU (torch.distributed.run): rank 0 owns
    the image, the optimizer and the regularizers, every rank evaluates tiles t with
    t mod world == rank on up to STREAMS_PER_GPU engines of its GPU."""

    def __init__(self, net, engines, rank, world, device, rows, cols, wire):
        import torch
        from style_transfer_amd import image_ops
        from style_transfer_amd.dist import DistributedTiles, broadcast_targets
        from style_transfer_amd.engine import DeviceArray
        from style_transfer_amd.farm import TileFarm, tile_grid
        from style_transfer_amd.optimizers import AdamOptimizer
        self.torch, self.image_ops = torch, image_ops
        self.net, self.engines, self.rank, self.world, self.device = net, engines, rank, world, device
        eng = self.eng = engines[0]
        self.H, self.W = rows * TILE, cols * TILE
        self.rects = tile_grid((self.H, self.W), TILE)
        self.tiles_per_rank = -(-len(self.rects) // world)
        self.wire = wire

        # ---- targets (once, outside the timed region): style Grams and the content map of the
        # image, computed on rank 0's GPU and broadcast device to device
        contents, styles = [], []
        if rank == 0:
            helper = TileFarm(net, verbose=False, engines=[eng])
            contents, styles = targets_on(helper, self.H, self.W)
        contents, styles = broadcast_targets(contents, styles, device)
        eng.set_contents_and_styles(contents, styles)          # (the rank's other engines share them)
        eng.sync()

        def wrap(tensor, engine):
            """A DeviceArray view of a torch tensor (no copy; torch keeps ownership)."""
            return DeviceArray.from_pointer(engine, tensor.data_ptr(), tensor.shape, owner=tensor)

        self.group_ms = []           # GPU time of one concurrent group of tile evaluations on this rank
        if rank == 0:
            self.img = eng.to_device(start_image(self.H, self.W))
            self.grad = eng.empty((3, self.H, self.W))
            self.old = eng.empty((3, self.H, self.W)).copy_from(self.img)
            self.rng = np.random.RandomState(0)
            self.opt = AdamOptimizer(eng, self.img, step_size=15, bp1=1 - 1 / 20, decay=0.05, power=0.5)
        inflight = []
        grad_bufs = [torch.empty((3, TILE, TILE), dtype=torch.float32, device=device)
                     for _ in range(self.tiles_per_rank)]
        tile_bufs = {}

        def evaluate_begin(jobs, roll):
            """Enqueues this rank's tiles, one per engine (they run concurrently)."""
            inflight.clear()
            for k, (tile, start) in enumerate(jobs):
                e = engines[k % len(engines)]
                inflight.append((e, k, e.sc_grad_tile_async(
                    wrap(tile, e), start, roll, CONTENT_LAYERS, STYLE_LAYERS, {}, CONTENT_WEIGHT,
                    STYLE_WEIGHT, grad_out=wrap(grad_bufs[k], e))))

        def evaluate_end():
            """Waits for them; [(loss, grad tensor)]."""
            used = []
            for e, _, _ in inflight:
                if e not in used:
                    used.append(e)
            for e in used:
                e.sync()
            per_engine = {}
            for e, _, _ in inflight:
                per_engine[id(e)] = per_engine.get(id(e), 0) + 1
            self.group_ms.append(max(e.last_tile_ms() * per_engine[id(e)] for e in used))
            return [(p.loss, grad_bufs[k]) for _, k, p in inflight]

        def cut(rect, roll):
            if rect not in tile_bufs:
                tile_bufs[rect] = torch.empty((3, rect[1] - rect[0], rect[3] - rect[2]),
                                              dtype=torch.float32, device=device)
            image_ops.cut_tile(eng, self.img, roll, rect, wrap(tile_bufs[rect], eng))
            return tile_bufs[rect]

        def put(rect, g, roll):
            image_ops.put_tile(eng, self.grad, roll, rect, wrap(g, eng))

        self.tiles = DistributedTiles(lambda rect, roll: (cut(rect, roll), eng.sync())[0],
                                      (evaluate_begin, evaluate_end), put, device)

    def step(self):
        if self.rank != 0:
            self.tiles.eval_sc_grad(self.rects, (0, 0))
            return None
        roll = draw_roll(self.rng, self.H, self.W)

        def opfunc(params):
            loss = self.tiles.eval_sc_grad(self.rects, roll)
            reg = self.image_ops.regularizers(self.eng, params, self.grad, MEAN, 5.0, 2.0, 2.0, 6.0)
            self.eng.sync()
            return loss + reg.value, self.grad
        avg, loss = self.opt.update(opfunc)
        self.image_ops.step_stats(self.eng, avg, self.old)
        return loss

    def fence(self):
        import torch.distributed as dist
        for e in self.engines:
            e.sync()
        self.torch.cuda.synchronize()
        dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup):
        """W untimed steps, then exactly K steps between two fences (barrier + device
        synchronisation on both sides); the MAX over ranks."""
        import torch.distributed as dist
        loss = None
        for _ in range(warmup):
            self.step()
        self.fence()
        self.group_ms.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = self.step()
        self.fence()
        elapsed = time.perf_counter() - t0
        t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.wire)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0]), loss

    def bit_identical(self):
        """One evaluation over all ranks against the same evaluation on rank 0's GPU alone (a
        collective call: every rank takes part in the first half).  True / False on rank 0."""
        from style_transfer_amd.farm import TileFarm
        roll = draw_roll(np.random.RandomState(123), self.H, self.W)
        loss_n = self.tiles.eval_sc_grad(self.rects, roll if self.rank == 0 else (0, 0))
        if self.rank != 0:
            return None
        self.eng.sync()
        grad_n = self.grad.get()
        solo = TileFarm(self.net, [self.eng.device], verbose=False, engines=list(self.engines))
        ref = self.eng.empty((3, self.H, self.W))
        loss_1 = solo.eval_sc_grad(self.img, ref, roll, CONTENT_LAYERS, STYLE_LAYERS, {},
                                   CONTENT_WEIGHT, STYLE_WEIGHT, TILE)
        same = bool(loss_n == loss_1 and np.array_equal(grad_n, ref.get()))
        ref.free()
        solo.close()
        return same


def bench_ranks(opts, net, rank, world, local_rank, device, rows, cols, debug_one_gpu):
    """N > 1: one process per GPU under torch.distributed.run.  Returns the line on rank 0."""
    import torch
    import torch.distributed as dist
    from style_transfer_amd import lib
    from style_transfer_amd.dist import broadcast_weights
    from style_transfer_amd.engine import TileEngine
    from style_transfer_amd.weights import synthetic_weights

    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo' if debug_one_gpu else 'nccl', rank=rank, world_size=world)
    # weights: built once on rank 0 and broadcast with ONE RCCL collective (80 MB); every other
    # rank sets its engines from the received device tensors
    weights = broadcast_weights(synthetic_weights(net, 0) if rank == 0 else None, device)
    # every rank: STREAMS_PER_GPU engines (HIP streams) on its GPU sharing one weight bank and one
    # target set
    engines = [TileEngine(net, local_rank, weights)]
    engines += [TileEngine(net, local_rank, share=engines[0]) for _ in range(STREAMS_PER_GPU - 1)]
    wire = 'cpu' if debug_one_gpu else device

    # ---- the headline: BASELINE's literal workload (strong scaling)
    job = RankJob(net, engines, rank, world, device, rows, cols, wire)
    same = job.bit_identical()
    elapsed, loss = job.timed(opts.steps, opts.warmup)
    timed_group_ms = list(job.group_ms)

    # ---- a second, longer measurement of the same loop (the timed region above is short)
    steady = None
    clock = None
    if opts.steady_seconds > 0:
        t = torch.tensor([max(1, int(np.ceil(opts.steady_seconds / (elapsed / opts.steps))))],
                         dtype=torch.int64, device=wire)
        dist.broadcast(t, 0)
        n_steady = int(t[0])
        if rank == 0:
            for e in engines:
                e.clock_marks(True)
        dt, _ = job.timed(n_steady, 0)
        if rank == 0:
            clock = clock_summary(engines)
        steady = {'steps': n_steady, 'seconds': dt, 'ms_per_step': dt / n_steady * 1e3,
                  'value': len(job.rects) * n_steady / dt, 'unit': 'tile-iterations/s'}
    line = None
    if rank == 0:
        line = base_line(opts, world, rows, cols, elapsed, loss, engines[0], timed_group_ms)
        line['layout'] = 'one process per GPU (torch.distributed.run), tiles and gradients as batched ' \
                         'point-to-point transfers over %s, no collective on the data path' \
                         % ('gloo through host memory (STX_BENCH_DEBUG_ONE_GPU: every rank on GPU 0)'
                            if debug_one_gpu else 'RCCL / xGMI')
        line['bit_identical'] = same
        line['peers_without_access'] = int(engines[0].query(lib.Q_PEERS_WITHOUT_ACCESS))
        add_clock(line['roofline'], clock)
        if steady is not None:
            line['steady'] = steady

    # ---- the weak-scaling sub-record (rounds 1-3's headline): four tiles per GPU, growing image
    if not opts.no_weak and not opts.debug_grid:
        wr, wc = WEAK_GRIDS[world]
        weak = RankJob(net, engines, rank, world, device, wr, wc, wire)
        weak_same = weak.bit_identical()
        w_elapsed, w_loss = weak.timed(opts.steps, min(opts.warmup, 2))
        if rank == 0:
            line['weak'] = {'scaling': 'weak', 'n_gpus': world,
                            'workload': 'VGG-19 %dx%d image, %d tiles of %dx%d per step, %d per GPU, -o adam'
                                        % (weak.W, weak.H, len(weak.rects), TILE, TILE, STREAMS_PER_GPU),
                            'value': len(weak.rects) * opts.steps / w_elapsed, 'unit': 'tile-iterations/s',
                            'ms_per_step': w_elapsed / opts.steps * 1e3, 'steps': opts.steps,
                            'final_loss': w_loss, 'avg_launch_ms': float(np.mean(weak.group_ms)),
                            'bit_identical': weak_same}
    # every rank lets go of its GPU before rank 0 goes on alone (farm and whole-run legs drive
    # all GPUs from one process); no collective is pending past this point
    dist.barrier()
    dist.destroy_process_group()
    for e in engines:
        e.close()
    return line


if __name__ == '__main__':
    main()
