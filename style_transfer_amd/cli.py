"""Command-line driver, flag-compatible with the reference's ``style_transfer.py`` CLI
(``style_transfer.py:1076-1163``): same options (``config_system.py``), the same per-step
console line and ``<RUN>_log.csv`` columns (``style_transfer.py:101-130,950-951``), the same
PNG comment block (``style_transfer.py:1003-1010``).  The web / GUI live views are not part of
the accelerated path; ``--display`` is accepted and ignored.
"""

from concurrent.futures import ThreadPoolExecutor
import csv
from datetime import datetime
import sys
import time

import numpy as np
from PIL import Image

from . import fastpng, image_ops, lib
from .config_system import parse_args
from .farm import TileFarm
from .netspec import load_net
from .transfer import StyleTransfer
from .weights import load_weights


class StatLogger:
    """Per-iteration rows for ``<RUN>_log.csv`` (style_transfer.py:101-130)."""

    def __init__(self):
        self.rows, self.start = [], None

    def new_row(self, **kw):
        if self.start is None:
            self.start = time.perf_counter()
        kw.update(iteration=len(self.rows), time=time.perf_counter() - self.start)
        self.rows.append(kw)

    def update(self, **kw):
        self.rows[-1].update(kw)

    def dump(self, path):
        fields = ['iteration', 'scale', 'step', 'time']
        for row in self.rows:
            fields += [k for k in row if k not in fields]
        with open(path, 'w', newline='') as f:
            writer = csv.DictWriter(f, fieldnames=fields)
            writer.writeheader()
            writer.writerows(self.rows)


class Progress:
    """Prints the reference's step line and keeps the statistics (style_transfer.py:912-960)."""

    def __init__(self, run, stats, save_every=0):
        self.run, self.stats, self.save_every = run, stats, save_every
        self.prev_t, self.step, self.steps = None, 0, 0
        self.writer = ThreadPoolExecutor(max_workers=1) if save_every else None
        self.writes = []

    MAX_PENDING_WRITES = 4      # frames (H x W x 3 bytes each) waiting for the encoder thread

    def finish(self, reraise=True):
        """Waits for the --save-every pictures still being encoded.  Writer errors are reported;
        they are raised only when no other exception is already on its way (``reraise``)."""
        errors = []
        for w in self.writes:
            try:
                w.result()
            except Exception as err:    # pylint: disable=broad-except
                errors.append(err)
        self.writes = []
        if self.writer is not None:
            self.writer.shutdown()
        for err in errors:
            print('--save-every: writing a picture failed: %s: %s' % (type(err).__name__, err),
                  file=sys.stderr)
        if errors and reraise:
            raise errors[0]

    def set_steps(self, steps):
        self.steps = steps

    def wants_image(self, n):
        """Will the callback for the n-th iteration of the run (1-based) read that iteration's
        image?  (StyleTransfer.transfer runs one iteration ahead of the GPU and calls back one
        iteration late -- except where the answer is yes.)"""
        return bool(self.save_every) and n % self.save_every == 0

    def __call__(self, step, update_size, loss, tv_loss, transfer):
        now = time.perf_counter()
        self.step += 1
        dt = 0 if self.prev_t is None else now - self.prev_t
        self.prev_t = now
        state = transfer.state
        self.stats.new_row(scale=state.scale, step=step - 1, content_h=transfer.img.shape[1],
                           content_w=transfer.img.shape[2])
        self.stats.update(update_size=update_size, loss=loss, tv_norm=tv_loss)
        if self.save_every and self.step % self.save_every == 0:
            # the pixels leave the GPU now; deflate and file I/O run beside the next steps
            rgb = image_ops.to_u8(transfer.engine, transfer.current_raw, transfer.mean)
            self.writes = [w for w in self.writes if not w.done() or w.exception() is not None]
            while sum(not w.done() for w in self.writes) >= self.MAX_PENDING_WRITES:
                next(w for w in self.writes if not w.done()).exception()     # (waits for it)
            self.writes.append(self.writer.submit(fastpng.save_rgb,
                                                  self.run + '_out_%04d.png' % self.step, rgb))
        print('Step %d, time: %.2f s, update: %.2f, loss: %e, tv: %.2f' %
              (step, dt, update_size, loss, tv_loss), flush=True)


def image_comment(args, argv):
    """The PNG iTXt comment (style_transfer.py:1003-1010): command line, then the option and
    run-state namespaces exactly as the reference prints them (``vars()`` of its lazy namespace,
    i.e. one ``ns: Namespace(...)`` and one ``state_obj: Namespace(...)`` line)."""
    s = 'Created with style_transfer_amd (CLI-compatible with crowsonkb/style_transfer).\n\n'
    s += 'Command line: style_transfer.py ' + ' '.join(argv) + '\n\nParameters:\n'
    for item in sorted(vars(args).items()):
        s += '%s: %s\n' % item
    return s


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    from argparse import Namespace
    state = Namespace()
    args = parse_args(state, argv)
    start_time = time.perf_counter()
    now = datetime.now()
    run = '%02d%02d%02d_%02d%02d%02d' % (now.year % 100, now.month, now.day, now.hour,
                                         now.minute, now.second)
    print('Run %s started.' % run)
    net = load_net(args.model)
    if args.list_layers:
        print('Layers:')
        for layer, shape in net.shapes().items():
            print('% 25s %s' % (layer, shape))
        return 0
    # Three things run side by side, each releasing the interpreter lock: the weight file is read
    # (or the seeded bank drawn) on one helper thread, the GPU runtime is woken up and the engines
    # (streams, scalar arenas) are created on another, and this thread decodes the pictures.
    if any(d < 0 for d in args.devices):
        # the reference runs device -1 as a Caffe-CPU worker (config_system.py:56,
        # style_transfer.py:193-201); there is no CPU path here, by design
        print('--devices -1 (CPU worker) is mapped to GPU 0: this engine has no CPU path.')
    devices = [d if d >= 0 else 0 for d in args.devices]

    def wake_gpus():
        if lib.device_count() < 1:
            raise RuntimeError('no AMD GPU visible: this engine has no CPU path')
        return TileFarm(net, devices, None)

    loader = ThreadPoolExecutor(max_workers=2)
    weights_future = loader.submit(load_weights, args.weights, net)
    farm_future = loader.submit(wake_gpus)
    try:
        print('Initializing %s on device(s) %s.' % (args.weights, devices))
        content_image = Image.open(args.content_image).convert('RGB')
        style_images = [Image.open(p).convert('RGB') for p in args.style_images]
        initial_image = Image.open(args.init_image).convert('RGB') if args.init_image else None
        aux_image = Image.open(args.aux_image).convert('RGB') if args.aux_image else None
        farm = farm_future.result()
        farm.set_weights(weights_future.result())
    finally:
        # (an error above must not wait for 80 MB of weights nobody will use)
        weights_future.cancel()
        loader.shutdown(wait=False)
    transfer = StyleTransfer(farm, args, state)
    stats = StatLogger()
    progress = Progress(run, stats, args.save_every)
    np.random.seed(args.seed)
    failed = True
    try:
        transfer.transfer_multiscale([content_image], style_images, initial_image, aux_image,
                                     callback=progress)
        failed = False
    except (EOFError, KeyboardInterrupt):
        print()
        failed = False
    finally:
        stats.dump(run + '_log.csv')
        progress.finish(reraise=not failed)
    if transfer.current_raw is not None:
        path = args.output_image or run + '_out.png'
        print('Saving output as %s.' % path)
        rgb = image_ops.to_u8(transfer.engine, transfer.current_raw, transfer.mean)
        comment = [('Comment', image_comment(args, argv))]
        if path.lower().endswith('.png'):
            # the reference's image.save(path, pnginfo=...) (style_transfer.py:1003-1010): same
            # pixels and comment, deflated on all cores instead of one
            fastpng.save_rgb(path, rgb, comment)
        else:
            Image.fromarray(rgb).save(path)
    spent = time.perf_counter() - start_time
    steps_time = sum(transfer.step_times)
    if steps_time > 0:
        print('%d tile-iterations in %.3f s of stepping: %.2f tile-iterations/s.' %
              (farm.tile_evals, steps_time, farm.tile_evals / steps_time))
    print('Run %s ending after %dm %.3fs.' % (run, spent // 60, spent % 60))
    farm.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
