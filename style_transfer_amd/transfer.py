"""The optimisation schedule: multi-scale pyramid, per-iteration seam-suppression shift, the
objective (tiles + regularizers) and the optimizer step.

Host-side restatement of ``StyleTransfer`` (``style_transfer.py:664-909``) and of the image
conversion helpers of ``CaffeModel`` (``style_transfer.py:378-401``) on top of the tile farm.
Differences in mechanism, not in result:
  * image, gradient and optimizer state live on the master GPU (``DeviceArray``);
  * the random shift of an iteration is NOT applied by rolling arrays: the image stays
    un-rolled and ``roll`` travels down to the tile cut / put kernels and to the engines'
    content-map addressing as an index offset;
  * the draw order of the global numpy RNG is the reference's, so a
    run with the same ``--seed`` visits the same shifts.
"""

from argparse import Namespace
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
import json
import sys
import time

import numpy as np
from PIL import Image

from . import image_ops, lib
from .config_system import ffloat
from .optimizers import AdamOptimizer, LBFGSOptimizer
from .resample import resample_device


def fit_size(wh, size, scale_up=False, div=1):
    """(w, h) of a picture fitted into a size x size square, None if it is left alone: the
    arithmetic of the reference's resize_to_fit (style_transfer.py:963-976) without the pixels,
    so that a whole run can be planned before anything is resampled."""
    size = int(round(size)) // div * div
    w, h = wh
    if not scale_up and max(w, h) <= size:
        return None
    if w > h:
        return size, int(round(size * h / w)) // div * div
    return int(round(size * w / h)) // div * div, size


def resize_to_fit(image, size, scale_up=False, div=1):
    """Resizes a PIL image to fit a size x size square (Lanczos), or returns it untouched."""
    target = fit_size(image.size, size, scale_up, div)
    return image if target is None else image.resize(target, Image.LANCZOS)


def pyramid_sizes(size, min_size):
    """[size, size/sqrt2, ...] down to min_size, largest first (style_transfer.py:840-846)."""
    sizes = [size]
    while True:
        size = round(size / np.sqrt(2))
        if size < min_size:
            return sizes
        sizes.append(size)


@dataclass
class ScalePlan:
    """One level of the pyramid, decided before the run starts."""
    index: int                  # 0 = coarsest
    size: int                   # the --size of this level
    content_wh: tuple           # picture size at this level
    style_fit: list             # per style image: (w, h) to resample to, or None = as is
    iterations: int
    tiles: tuple                # (rows, columns) of the tile grid (style_transfer.py:619-632)


def plan_scales(args, content_wh, style_whs):
    """The whole multi-scale schedule as data (style_transfer.py:832-909): pyramid sizes from
    --min-size up to --size, the picture size of every level, how each style image is fitted
    (--style-scale / --max-style-size / --style-scale-up; --style-multiscale keeps the
    originals and fits them in preprocess_images instead), the iteration count (the last
    --iterations value repeats) and the tile grid."""
    plans = []
    for i, size in enumerate(reversed(pyramid_sizes(args.size, args.min_size))):
        cw, ch = fit_size(content_wh, size, scale_up=True, div=args.div)
        fits = []
        for wh in style_whs:
            if args.style_multiscale:
                fits.append(None)
            elif args.style_scale >= 32:
                fits.append(fit_size(wh, args.style_scale, scale_up=True, div=args.div))
            else:
                target = round(size * args.style_scale)
                if args.max_style_size is not None:
                    target = min(target, args.max_style_size)
                fits.append(fit_size(wh, target, scale_up=args.style_scale_up, div=args.div))
        plans.append(ScalePlan(i, size, (cw, ch), fits,
                               args.iterations[min(i, len(args.iterations) - 1)],
                               ((ch - 1) // args.tile_size + 1, (cw - 1) // args.tile_size + 1)))
    return plans


def style_pyramid(args):
    """Sizes at which every style image is processed: [None] (as given) unless
    --style-multiscale MIN MAX asks for a sqrt(2) ladder (style_transfer.py:493-503)."""
    if not args.style_multiscale:
        return [None]
    vmin, vmax = args.style_multiscale
    sizes = [vmax]
    while True:
        nxt = int(round(sizes[-1] / np.sqrt(2)))
        if nxt < max(32, vmin):
            return sizes
        sizes.append(nxt)


def parse_weights(args, master_weight):
    """['name', 'name:2', ...] -> (names, {name: weight normalised to sum |w| = master})
    (style_transfer.py:684-698)."""
    names, weights, total = [], {}, 0
    for arg in args:
        name, _, w = arg.partition(':')
        names.append(name)
        weights[name] = ffloat(w) if w else 1
        total += abs(weights[name])
    return names, {n: w * master_weight / total for n, w in weights.items()}


class StyleTransfer:
    """Runs style transfer on a ``TileFarm``.  ``args`` is the option namespace of
    ``config_system.parse_args`` (any object with the same attributes works)."""

    def __init__(self, farm, args, state=None):
        self.farm = farm
        self.engine = farm.master
        self.args = args
        self.state = state if state is not None else Namespace()
        self.mean = np.float32(args.mean).reshape((3, 1, 1))
        self.layer_weights = {layer: 1.0 for layer in farm.layers() + ['data']}
        if args.layer_weights:
            with open(args.layer_weights) as f:
                self.layer_weights.update(json.load(f))
        self.contents, self.styles = [], []
        self.img = None             # DeviceArray [3,H,W]: the raw iterate (params)
        self.grad = None
        self.old_avg = None
        self.aux_image = None
        self.optimizer = None
        self.current_raw = None     # DeviceArray: averaged iterate of the last step
        self.step = 0
        self.step_times = []
        # calls made to the callback object at hand (transfer's run-ahead loop asks it about its n-th call)
        self._cb_calls = [None, 0]
        self._converted = {}        # id(PIL image) -> (image, float array), filled by the helper thread
        # --swt-weight (style_transfer.py:716-720) calls PyWavelets, which is not part of the
        # reference tree; its transform is restated for the command line's defaults only
        raw = getattr(getattr(args, 'ns', args), 'swt_weight', 0)
        if (callable(raw) or raw) and (str(args.swt_wavelet) not in ('haar', 'db1') or
                                       int(args.swt_levels) != 1):
            raise NotImplementedError('--swt-wavelet %s --swt-levels %s: only the default '
                                      '(haar, 1 level) is implemented'
                                      % (args.swt_wavelet, args.swt_levels))
        if callable(raw) or raw:
            import warnings
            warnings.warn('--swt-weight: the Haar SWT term is a restatement of PyWavelets\' '
                          'swt2 / iswt2, which is not part of the reference tree and not '
                          'installed here; it has never been compared with it (parity unpinned)')

    # ----------------------------------------------------------------------- image <-> params
    def pil_to_image(self, img):
        """RGB PIL image -> BGR CHW float32 minus mean (style_transfer.py:388-393).  The pictures
        of the next pyramid level are converted ahead of time on the helper thread that resizes
        them (transfer_multiscale): looked up by identity here."""
        cached = self._converted.get(id(img))
        if cached is not None and cached[0] is img:
            return cached[1]
        arr = np.float32(img).transpose((2, 0, 1))[::-1]
        return np.ascontiguousarray(arr - self.mean)

    def get_image(self, params=None):
        """PIL image of a device iterate (style_transfer.py:378-386)."""
        params = self.current_raw if params is None else params
        return Image.fromarray(image_ops.to_u8(self.engine, params, self.mean))

    @property
    def current_output(self):
        return self.get_image() if self.current_raw is not None else None

    # -------------------------------------------------------------------------- preprocessing
    def _style_variants(self, index, image):
        """The resamplings of one style image that contribute a Gram each: smallest ladder size
        first, stopping after the first one that no longer shrinks the picture; variants under
        32 pixels are skipped (style_transfer.py:505-531)."""
        for size in reversed(style_pyramid(self.args)):
            if size is None:
                yield image
                return
            scaled = resize_to_fit(image, size, div=self.args.div)
            last = max(scaled.size) == max(image.size)
            if min(scaled.size) >= 32:
                print('Processing style {} at {}x{}.'.format(index + 1, *scaled.size))
                yield scaled
            if last:
                return

    def preprocess_images(self, content_images, style_images, content_layers, style_layers,
                          roll=None):
        """Targets of one scale: the style Grams, averaged with equal weight over every style
        image and ladder size, and the tiling-averaged content features
        (style_transfer.py:488-554).  Everything stays on the master GPU.  ``roll`` (--jitter,
        once per iteration): features of the pictures rolled by it, one pass, no messages."""
        farm, tile = self.farm, self.args.tile_size
        if roll is None:
            print('Preprocessing the style image(s)...')
        if not self.styles:
            total, count = {}, 0
            for index, image in enumerate(style_images):
                for variant in self._style_variants(index, image):
                    feats = farm.prepare_features_device(self.pil_to_image(variant), style_layers,
                                                         tile, passes=1, roll=roll)
                    for layer, feat in feats.items():
                        gram = farm.gram_matrix(feat)
                        feat.free()
                        total[layer] = gram if layer not in total else total[layer] + gram
                    count += 1
            self.styles.append({layer: gram / count for layer, gram in total.items()})
        if roll is None:
            print('Preprocessing the content image(s)...')
        self.contents += [farm.prepare_features_device(self.pil_to_image(image), content_layers,
                                                       tile, passes=10 if roll is None else 1,
                                                       roll=roll)
                          for image in content_images]

    def _drop_contents(self):
        for content in self.contents:
            for feat in content.values():
                if hasattr(feat, 'free'):
                    feat.free()
        self.contents = []

    # ------------------------------------------------------------------------------ objective
    def eval_loss_and_grad(self, params, sc_args):
        """Loss and gradient of the full image (style_transfer.py:700-736).  ``params`` is the
        device iterate; returns (loss, device gradient).  Nothing waits on the host here: the
        loss is a LazyLoss (``float(loss)`` waits for it) and the gradient is complete in stream
        order on the master GPU, which is all the optimizers need -- the step loop synchronises
        once per iteration, for its statistics."""
        args = self.args
        (roll, content_layers, style_layers, content_weight, style_weight, dd_layers, dd_weight,
         content_roll) = sc_args
        lw = self.layer_weights['data']
        loss = self.farm.eval_sc_grad(params, self.grad, roll, content_layers, style_layers,
                                      self.layer_weights, content_weight, style_weight,
                                      args.tile_size, dd_layers=dd_layers, dd_weight=dd_weight,
                                      content_roll=content_roll, lazy=True)
        aux_on = self.aux_image is not None
        if args.tv_weight or args.p_weight or aux_on:
            reg = image_ops.regularizers(
                self.engine, params, self.grad, self.mean, lw * args.tv_weight, args.tv_power,
                lw * args.p_weight, args.p_power, self.aux_image,
                lw * args.aux_weight if aux_on else 0.0, aux_roll=roll)
            loss.add(reg, self.engine)
        if args.swt_weight:
            # style_transfer.py:716-720.  Only the reference's default transform is restated
            # (oracle/num_ops.py): PyWavelets, which it calls, is not part of its tree.
            swt = image_ops.swt_haar(self.engine, params, self.grad, lw * args.swt_weight,
                                     args.swt_power, roll=roll)
            loss.add(swt, self.engine)
        return loss, self.grad

    # --------------------------------------------------------------------------- one scale
    def transfer(self, iterations, content_images, style_images, callback=None):
        """Optimises the current image for ``iterations`` steps at the current scale
        (style_transfer.py:738-830)."""
        args, state = self.args, self.state
        state.scale = state.scale + 1 if 'scale' in state else 0
        state.step, state.steps = 0, iterations
        state.img_size = self.img.shape[1:]

        content_layers, content_weight = parse_weights(args.content_layers, args.content_weight)
        style_layers, style_weight = parse_weights(args.style_layers, 1)
        dd_layers, dd_weight = parse_weights(args.dd_layers, args.dd_weight)
        jitter = bool(args.jitter)
        self._drop_contents()                  # device-resident maps of the previous scale
        if not args.style_multiscale:
            self.styles = []
        # --jitter: the content maps are recomputed every iteration from the shifted picture
        # (style_transfer.py:757-763,789-794), only the style targets are fixed per scale
        self.preprocess_images([] if jitter else content_images, style_images,
                               [] if jitter else content_layers, style_layers)
        self.farm.set_contents_and_styles(self.contents, self.styles)

        if self.grad is None or self.grad.shape != self.img.shape:
            for buf in (self.grad, self.old_avg):
                if buf is not None:
                    buf.free()
            self.grad = self.engine.empty(self.img.shape)
            self.old_avg = self.engine.empty(self.img.shape)
        self.old_avg.copy_from(self.img)
        self.step += 1
        deepest_content = [l for l in reversed(self.farm.layers()) if l in content_layers][0]
        jitter_scale, _ = self.farm.layer_info(deepest_content)
        img_size = np.array(self.img.shape[-2:])

        # The host runs ONE iteration ahead of the GPU where it can: iteration i + 1 is queued
        # before the loss and statistics of iteration i are collected (a fence per iteration,
        # LazyLoss.seal), so the GPU does not idle while those travel home and the host queues
        # the next ~350 launches.  The reference blocks on every iteration (style_transfer.py:
        # 799-815); the values and their order are the same.  Not with L-BFGS (its curvature test
        # needs a dot product on the host inside the step), not with --jitter (targets change
        # every iteration), not with a callback this loop knows nothing about (it may read the
        # image of the step it is called for: only callbacks with ``wants_image`` are run behind).
        run_ahead = (args.optimizer == 'adam' and not jitter and
                     (callback is None or hasattr(callback, 'wants_image')))
        in_flight = []
        t_prev = [time.perf_counter()]
        # `wants_image(n)` is asked about the n-th call THIS callback object gets (1-based, over all
        # scales of all runs it is passed to through this StyleTransfer): counted per object, so that
        # a fresh callback on a reused StyleTransfer starts at 1 again
        if self._cb_calls[0] is not callback:
            self._cb_calls = [callback, 0]

        def finish(item):
            step_i, loss_i, stats_i = item
            loss_v = float(loss_i)          # waits for that iteration's fences only
            update_size, tv_loss = stats_i.values()
            now = time.perf_counter()
            self.step_times.append(now - t_prev[0])
            t_prev[0] = now
            self._cb_calls[1] += 1
            if callback is not None:
                callback(step=step_i, update_size=update_size, loss=loss_v, tv_loss=tv_loss,
                         transfer=self)

        try:
            self._step_loop(iterations, callback, run_ahead, in_flight, finish, jitter, jitter_scale,
                            img_size, state, content_images, content_layers, style_layers,
                            content_weight, style_weight, dd_layers, dd_weight)
        except BaseException as exc:
            # an interrupt or a host-side error in iteration i + 1 must not lose the finished iteration i
            # (its statistics row, its --save-every picture) -- but when the engine itself failed there is
            # nothing to collect, and whatever the drain raises is reported beside the original error,
            # never swallowed; the engines are left idle either way
            if not isinstance(exc, lib.StxError):
                while in_flight:
                    item = in_flight.pop(0)
                    try:
                        finish(item)
                    except Exception as drain_exc:      # pylint: disable=broad-except
                        print('style_transfer_amd: while collecting step %d after %r: %r'
                              % (item[0], exc, drain_exc), file=sys.stderr)
                        break
            del in_flight[:]
            try:
                for eng in getattr(self.farm, 'engines', ()):
                    eng.sync()
            except Exception as sync_exc:               # pylint: disable=broad-except
                print('style_transfer_amd: engines not idle after %r: %r' % (exc, sync_exc), file=sys.stderr)
            raise
        while in_flight:
            finish(in_flight.pop(0))
        return self.current_raw

    def _step_loop(self, iterations, callback, run_ahead, in_flight, finish, jitter, jitter_scale, img_size,
                   state, content_images, content_layers, style_layers, content_weight, style_weight,
                   dd_layers, dd_weight):
        for step in range(1, iterations + 1):
            t0 = time.perf_counter()
            state.step = step - 1
            # the iteration's random shift (style_transfer.py:777-786); the reference rolls the
            # image and the optimizer state by xy * jitter_scale, here it is an index offset
            scale = 1 if jitter else jitter_scale
            xy = np.int32(np.random.uniform(-0.5, 0.5, size=2) * img_size) // scale
            roll = xy * scale
            self.optimizer.roll(roll)
            content_roll = None
            if jitter:
                # any pixel shift; the engines get content maps of the shifted picture and leave
                # them where they are
                self._drop_contents()
                self.preprocess_images(content_images, [], content_layers, [], roll=roll)
                self.farm.set_contents_and_styles(self.contents, self.styles)
                content_roll = (0, 0)
            sc_args = (roll, content_layers, style_layers, content_weight, style_weight,
                       dd_layers, dd_weight, content_roll)
            avg_img, loss = self.optimizer.update(lambda p: self.eval_loss_and_grad(p, sc_args))
            self.optimizer.roll(-roll)
            if run_ahead:
                stats = image_ops.step_stats_async(self.engine, avg_img, self.old_avg)
                loss.seal(also=[self.engine])
                self.current_raw = avg_img
                if in_flight:
                    finish(in_flight.pop())
                in_flight.append((step, loss, stats))
                # a callback that will look at this step's image gets it before the next step
                # overwrites it
                if step == iterations or (callback is not None and
                                          callback.wants_image(self._cb_calls[1] + len(in_flight))):
                    finish(in_flight.pop())
                continue
            update_size, tv_loss = image_ops.step_stats(self.engine, avg_img, self.old_avg)
            loss = float(loss)              # (everything is finished by now: publishes the terms)
            self.current_raw = avg_img
            self.step_times.append(time.perf_counter() - t0)
            self._cb_calls[1] += 1
            if callback is not None:
                callback(step=step, update_size=update_size, loss=loss, tv_loss=tv_loss,
                         transfer=self)

    # ------------------------------------------------------------------------- all scales
    def _first_iterate(self, plan, initial_image):
        """The start image of the coarsest level and its optimizer (style_transfer.py:883-900):
        the supplied --init-image, or uniform noise drawn from the global RNG."""
        args = self.args
        w, h = plan.content_wh
        if initial_image:
            start = self.pil_to_image(initial_image.resize((w, h), Image.LANCZOS))
        else:
            start = self.pil_to_image(np.random.uniform(0, 255, size=(h, w, 3)))
        self.img = self.engine.to_device(start)
        if args.optimizer == 'adam':
            # an image that is already a picture starts with a biased first moment
            return AdamOptimizer(self.engine, self.img, step_size=args.step_size,
                                 bp1=1 - (1 / args.avg_window), decay=args.step_decay[0],
                                 power=args.step_decay[1], biased_g1=bool(initial_image))
        if args.optimizer == 'lbfgs':
            return LBFGSOptimizer(self.engine, self.img)
        raise ValueError(args.optimizer)

    def transfer_multiscale(self, content_images, style_images, initial_image=None, aux_image=None,
                            callback=None):
        """Runs the planned pyramid (plan_scales), coarsest level first; every level starts from
        the Lanczos-upsampled averaged iterate of the one before (style_transfer.py:832-909)."""
        args = self.args
        if any(image.size != content_images[0].size for image in content_images):
            raise ValueError('All of the content images must be the same size')
        plans = plan_scales(args, content_images[0].size, [image.size for image in style_images])
        if callback is not None and hasattr(callback, 'set_steps'):
            callback.set_steps(sum(plan.iterations for plan in plans))
        def resized(plan):
            # the content / style pictures of one level (style_transfer.py:856-868), resized and
            # already converted to the network's input format (0.15 s per run on the main thread)
            w, h = plan.content_wh
            contents = [image.resize((w, h), Image.LANCZOS) for image in content_images]
            styles = [image if fit is None else image.resize(fit, Image.LANCZOS)
                      for image, fit in zip(style_images, plan.style_fit)]
            converted = {}
            for image in contents + styles:
                arr = np.float32(image).transpose((2, 0, 1))[::-1]
                converted[id(image)] = (image, np.ascontiguousarray(arr - self.mean))
            return contents, styles, converted

        # The pictures of the next level are resized on a helper thread while the GPU works on
        # the current one (Pillow releases the interpreter lock inside resize; the main thread
        # sits in stx_sync most of the time): 0.1 s per 4096-pixel picture and level otherwise.
        pool = ThreadPoolExecutor(max_workers=1)
        try:
            pending = pool.submit(resized, plans[0]) if plans else None
            previous = None
            for number, plan in enumerate(plans):
                w, h = plan.content_wh
                print('\nScale %d, image size %dx%d.\n' % (plan.index + 1, w, h))
                contents, styles, self._converted = pending.result()
                pending = pool.submit(resized, plans[number + 1]) if number + 1 < len(plans) else None
                if aux_image:
                    if self.aux_image is not None:
                        self.aux_image.free()
                    self.aux_image = self.engine.to_device(
                        self.pil_to_image(aux_image.resize((w, h), Image.LANCZOS)))
                if previous is None:
                    self.optimizer = self._first_iterate(plan, initial_image)
                else:
                    # model.resize_image (style_transfer.py:399-401): Lanczos, on the GPU
                    self.img = resample_device(self.engine, previous, (h, w))
                    self.optimizer.set_params(self.img)
                previous = self.transfer(plan.iterations, contents, styles, callback)
        finally:
            pool.shutdown()
        return self.current_output
