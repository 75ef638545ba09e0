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

N > 1 is weak scaling: every rank evaluates four 1024 x 1024 tiles per step and the image grows
with N (2048 x 4096, 4096 x 4096 -- config 4's top scale -- and 4096 x 8192 at N = 2, 4, 8); rank 0
owns the image and the optimizer, tiles go out and gradients come back as batched point-to-point
transfers over RCCL, there is no collective on the data path.  value = tile-iterations per second
of the whole job.

The line also carries
  roofline      algorithmic FLOP of the four tile-iterations of one GPU (SURVEY.md section 8d:
                1 514 240 FLOP per tile pixel) over the GPU time of that concurrent group of
                stx_sc_grad_tile calls -- HIP events on each engine's own stream inside the timed
                region, the longest of the four spans -- against the fp32 MFMA peak;
  cpu_baseline  the numpy oracle (a port of the reference's Caffe-CPU path) timed on this box's
                host cores on a bounded sample -- rank 0, N = 1 only.
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


def measured_traffic():
    """HBM bytes per stx_sc_grad_tile launch from the PMC passes of this build
    (profiles/r01_i_hbm_traffic_pmc.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
    separate runs of this script, FETCH_SIZE doubled per the gfx950 calibration on the Adam
    kernel -- tools/pmc_traffic.py).  None if the file is missing."""
    path = os.path.join(REPO, 'profiles', 'r01_i_hbm_traffic_pmc.json')
    try:
        with open(path) as f:       # measured per tile-iteration; one launch group = 4 of them
            return float(json.load(f)['hbm_bytes_per_tile_iteration']) * TILES_PER_GPU
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(net):
    """Times the oracle's tile evaluation on the host cores (checker used as a yardstick only)."""
    from oracle.caffe_net import synthetic_weights
    from oracle.tile_path import OracleModel
    size = 512
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
    while reps < 2 or (time.perf_counter() - t0 < 10 and reps < 8):
        om.sc_grad_tile(tile, (0, 0), CONTENT_LAYERS, STYLE_LAYERS, {}, cw, sw)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    return {'value': 1.0 / (dt * (TILE * TILE) / (size * size)), 'unit': 'tile-iterations/s',
            'cores': cores, 'kind': 'port',
            'sample': '%d VGG-19 tile-iterations at %dx%d with the numpy oracle (im2col + '
                      'multithreaded SGEMM, %.2f s each), scaled by pixel count to the '
                      '%dx%d benchmark tile' % (reps, size, size, dt, TILE, TILE)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    opts = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo' if debug_one_gpu else 'nccl', rank=rank, world_size=world)

    from style_transfer_amd import image_ops
    from style_transfer_amd.engine import DeviceArray, TileEngine
    from style_transfer_amd.farm import TileFarm, tile_grid
    from style_transfer_amd.netspec import builtin_net
    from style_transfer_amd.optimizers import AdamOptimizer
    from style_transfer_amd.weights import synthetic_weights

    net = builtin_net('vgg19')
    weights = synthetic_weights(net, 0)
    # every rank: TILES_PER_GPU engines (HIP streams) on its GPU, one tile of the step each
    engines = [TileEngine(net, local_rank, weights) for _ in range(TILES_PER_GPU)]
    eng = engines[0]
    rows, cols = GRIDS[world]
    H, W = rows * TILE, cols * TILE
    rects = tile_grid((H, W), TILE)
    assert len(rects) == TILES_PER_GPU * world
    content_weight = {'conv4_2': 0.05}
    style_weight = {l: 1.0 / len(STYLE_LAYERS) for l in STYLE_LAYERS}

    # ---- targets (once, outside the timed region): style Grams and the content map of the image
    contents, styles = [], []
    if rank == 0:
        helper = TileFarm(net, verbose=False, engines=[eng])
        style_feats = helper.prepare_features_device(smooth_picture(7, TILE, TILE), STYLE_LAYERS,
                                                     TILE, passes=1)
        styles = [{l: eng.gram_matrix(f) for l, f in style_feats.items()}]
        contents = [{l: f.get() for l, f in helper.prepare_features_device(
            smooth_picture(8, H, W), CONTENT_LAYERS, TILE, passes=1).items()}]
    if world > 1:
        from style_transfer_amd.dist import DistributedTiles, broadcast_targets
        contents, styles = broadcast_targets(contents, styles, device)
    for e in engines:
        e.set_contents_and_styles(contents, styles)

    def wrap(tensor, engine):
        """A DeviceArray view of a torch tensor (no copy; torch keeps ownership)."""
        arr = DeviceArray.__new__(DeviceArray)
        arr.engine, arr.shape, arr.dtype = engine, tuple(tensor.shape), np.dtype(np.float32)
        arr.nbytes, arr.ptr = tensor.numel() * 4, tensor.data_ptr()
        arr.free = lambda: None
        return arr

    group_ms = []           # GPU time of one concurrent group of tile evaluations on this rank
    state = {}
    if rank == 0:
        rng = np.random.RandomState(0)
        # the reference's start image: uniform noise minus the mean (style_transfer.py:889)
        img = eng.to_device(rng.uniform(0, 255, (3, H, W)).astype(np.float32) -
                            np.float32(MEAN).reshape(3, 1, 1))
        state.update(img=img, grad=eng.empty((3, H, W)),
                     old=eng.empty((3, H, W)).copy_from(img), rng=np.random.RandomState(0),
                     opt=AdamOptimizer(eng, img, step_size=15, bp1=1 - 1 / 20, decay=0.05,
                                       power=0.5))

    inflight = []

    def evaluate_begin(jobs, roll):
        """Enqueues this rank's tiles, one per engine (they run concurrently)."""
        inflight.clear()
        for k, (tile, start) in enumerate(jobs):
            e = engines[k % len(engines)]
            g = grad_bufs[k]
            inflight.append((e, k, e.sc_grad_tile_async(
                wrap(tile, e), start, roll, CONTENT_LAYERS, STYLE_LAYERS, {}, content_weight,
                style_weight, grad_out=wrap(g, e))))

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

    def evaluate(jobs, roll):
        evaluate_begin(jobs, roll)
        return evaluate_end()

    grad_bufs = [torch.empty((3, TILE, TILE), dtype=torch.float32, device=device)
                 for _ in range(TILES_PER_GPU)]
    tile_bufs = {}

    def cut(rect, roll):
        if rect not in tile_bufs:
            tile_bufs[rect] = torch.empty((3, rect[1] - rect[0], rect[3] - rect[2]),
                                          dtype=torch.float32, device=device)
        image_ops.cut_tile(eng, state['img'], roll, rect, wrap(tile_bufs[rect], eng))
        return tile_bufs[rect]

    def put(rect, g, roll):
        image_ops.put_tile(eng, state['grad'], roll, rect, wrap(g, eng))

    if world > 1:
        farm = DistributedTiles(lambda rect, roll: (cut(rect, roll), eng.sync())[0],
                                (evaluate_begin, evaluate_end), put, device)

    def eval_sc_grad(roll):
        if world > 1:
            return farm.eval_sc_grad(rects, roll)
        tiles = [cut(rect, roll) for rect in rects]
        eng.sync()
        results = evaluate([(t, (r[0], r[2])) for t, r in zip(tiles, rects)], roll)
        for rect, (_, g) in zip(rects, results):
            put(rect, g, roll)
        return sum(l for l, _ in results)

    def step():
        if rank != 0:
            farm.eval_sc_grad(rects, (0, 0))
            return None
        xy = np.int32(state['rng'].uniform(-0.5, 0.5, size=2) * (H, W)) // 8
        roll = xy * 8

        def opfunc(params):
            loss = eval_sc_grad(roll)
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
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

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
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if debug_one_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    if rank == 0:
        ms_per_step = elapsed / opts.steps * 1e3
        tiles_per_step = len(rects)
        avg_group_ms = float(np.mean(group_ms))
        flop = FLOP_PER_TILE_PIXEL * TILE * TILE * TILES_PER_GPU
        achieved = flop / (avg_group_ms * 1e-3) / 1e12
        # what the kernels actually put on the matrix cores: the 3x3 layers run Winograd kernels
        # that issue 4/9 (2-D) or 2/3 (1-D) of the direct-convolution MFMAs
        conv_alg, conv_issued = eng.last_tile_flops()
        issued = (flop / TILES_PER_GPU - conv_alg + conv_issued) * TILES_PER_GPU
        issued_tflops = issued / (avg_group_ms * 1e-3) / 1e12
        line = {
            'metric': 'tile-iterations/sec, VGG-19 2048px/1024-tile (fwd+bwd, Gram/content losses, '
                      'regularizers, Adam step)',
            'value': tiles_per_step * opts.steps / elapsed,
            'unit': 'tile-iterations/s',
            'n_gpus': world, 'steps': opts.steps, 'warmup': opts.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'VGG-19 --size %d --tile-size %d -o adam: %dx%d image, %d tiles of '
                                   '%dx%d per step, %d per GPU' % (max(H, W), TILE, W, H, tiles_per_step,
                                                                 TILE, TILE, TILES_PER_GPU),
                       'content_layers': CONTENT_LAYERS, 'style_layers': STYLE_LAYERS,
                       'tiles_per_step': tiles_per_step, 'tiles_per_gpu': TILES_PER_GPU,
                       'final_loss': loss},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
                         'traffic': measured_traffic(), 'traffic_unit': 'bytes per launch',
                         'kernel': 'stx_sc_grad_tile x %d concurrent on one GPU (conv_wino2_kernel '
                                   'fwd/dgrad, conv_mfma_kernel first layer + SYMM, gram)' % TILES_PER_GPU,
                         'flop_per_launch': flop, 'avg_launch_ms': avg_group_ms,
                         'mfma_issued': issued_tflops,
                         'mfma_issued_frac': issued_tflops / PEAK_FP32_MFMA_TFLOPS,
                         'note': 'achieved / frac count every convolution as a direct one (SURVEY 8d: '
                                 '1 514 240 FLOP per tile pixel); the 3x3 layers run Winograd '
                                 'F(2x2,3x3) kernels that issue 4/9 of those MFMAs, so frac can exceed '
                                 '1 -- mfma_issued(_frac) is the matrix-core work actually issued, '
                                 'against the same fp32 MFMA peak'},
        }
        if world == 1 and not opts.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(net)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for e in engines:
        e.close()


if __name__ == '__main__':
    main()
