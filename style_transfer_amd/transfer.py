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
import json
import time

import numpy as np
from PIL import Image

from . import image_ops
from .config_system import ffloat
from .optimizers import AdamOptimizer, LBFGSOptimizer
from .resample import resample_device


def resize_to_fit(image, size, scale_up=False, div=1):
    """Resizes a PIL image to fit a size x size square (style_transfer.py:963-976)."""
    size = int(round(size)) // div * div
    w, h = image.size
    if not scale_up and max(w, h) <= size:
        return image
    if w > h:
        new_w, new_h = size, int(round(size * h / w)) // div * div
    else:
        new_h, new_w = size, int(round(size * w / h)) // div * div
    return image.resize((new_w, new_h), Image.LANCZOS)


def pyramid_sizes(size, min_size):
    """[size, size/sqrt2, ...] down to min_size, largest first (style_transfer.py:840-846)."""
    sizes = [size]
    while True:
        size = round(size / np.sqrt(2))
        if size < min_size:
            return sizes
        sizes.append(size)


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
        for name in ('swt_weight', 'dd_weight', 'jitter'):
            # lazy (callable) values could turn non-zero mid-run: refuse them outright
            raw = getattr(getattr(args, 'ns', args), name, 0)
            if callable(raw) or raw:
                raise NotImplementedError('--%s is outside the accelerated path '
                                          '(reference default is off)' % name.replace('_', '-'))

    # ----------------------------------------------------------------------- image <-> params
    def pil_to_image(self, img):
        """RGB PIL image -> BGR CHW float32 minus mean (style_transfer.py:388-393)."""
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
    def preprocess_images(self, content_images, style_images, content_layers, style_layers):
        """Style Grams (mean over all style images and scales) and tiling-averaged content
        features (style_transfer.py:488-554)."""
        args, farm = self.args, self.farm
        print('Preprocessing the style image(s)...')
        sizes = [None]
        if args.style_multiscale:
            vmin, vmax = args.style_multiscale
            size, sizes = vmax, [vmax]
            while True:
                size = int(round(size / np.sqrt(2)))
                if size < max(32, vmin):
                    break
                sizes.append(size)
        if not self.styles:
            grams, count = {}, 0
            for i, image in enumerate(style_images):
                too_big = False
                for size in reversed(sizes):
                    if too_big:
                        break
                    if size:
                        scaled = resize_to_fit(image, size, div=args.div)
                        if max(scaled.size) == max(image.size):
                            too_big = True
                        if min(scaled.size) < 32:
                            continue
                        print('Processing style {} at {}x{}.'.format(i + 1, *scaled.size))
                    else:
                        scaled = image
                    feats = farm.prepare_features_device(self.pil_to_image(scaled), style_layers,
                                                         args.tile_size, passes=1)
                    for layer, feat in feats.items():
                        gram = farm.gram_matrix(feat)
                        feat.free()
                        grams[layer] = gram if layer not in grams else grams[layer] + gram
                    count += 1
            for gram in grams.values():
                gram /= count
            self.styles.append(grams)
        print('Preprocessing the content image(s)...')
        for image in content_images:
            self.contents.append(farm.prepare_features_device(
                self.pil_to_image(image), content_layers, args.tile_size, passes=10))

    # ------------------------------------------------------------------------------ objective
    def eval_loss_and_grad(self, params, sc_args):
        """Loss and gradient of the full image (style_transfer.py:700-736).  ``params`` is the
        device iterate; returns (loss, device gradient)."""
        args = self.args
        roll, content_layers, style_layers, content_weight, style_weight = sc_args
        lw = self.layer_weights['data']
        loss = self.farm.eval_sc_grad(params, self.grad, roll, content_layers, style_layers,
                                      self.layer_weights, content_weight, style_weight,
                                      args.tile_size)
        aux_on = self.aux_image is not None
        if args.tv_weight or args.p_weight or aux_on:
            reg = image_ops.regularizers(
                self.engine, params, self.grad, self.mean, lw * args.tv_weight, args.tv_power,
                lw * args.p_weight, args.p_power, self.aux_image,
                lw * args.aux_weight if aux_on else 0.0, aux_roll=roll)
            self.engine.sync()
            loss += reg.value
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
        for content in self.contents:          # device-resident maps of the previous scale
            for feat in content.values():
                if hasattr(feat, 'free'):
                    feat.free()
        self.contents = []
        if not args.style_multiscale:
            self.styles = []
        self.preprocess_images(content_images, style_images, content_layers, style_layers)
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

        for step in range(1, iterations + 1):
            t0 = time.perf_counter()
            state.step = step - 1
            # the iteration's random shift (style_transfer.py:777-786); the reference rolls the
            # image and the optimizer state by xy * jitter_scale, here it is an index offset
            xy = np.int32(np.random.uniform(-0.5, 0.5, size=2) * img_size) // jitter_scale
            roll = xy * jitter_scale
            self.optimizer.roll(roll)
            sc_args = (roll, content_layers, style_layers, content_weight, style_weight)
            avg_img, loss = self.optimizer.update(lambda p: self.eval_loss_and_grad(p, sc_args))
            self.optimizer.roll(-roll)
            update_size, tv_loss = image_ops.step_stats(self.engine, avg_img, self.old_avg)
            self.current_raw = avg_img
            self.step_times.append(time.perf_counter() - t0)
            if callback is not None:
                callback(step=step, update_size=update_size, loss=loss, tv_loss=tv_loss,
                         transfer=self)
        return self.current_raw

    # ------------------------------------------------------------------------- all scales
    def transfer_multiscale(self, content_images, style_images, initial_image=None, aux_image=None,
                            callback=None):
        """The sqrt(2) pyramid from --min-size up to --size (style_transfer.py:832-909)."""
        args = self.args
        sizes = pyramid_sizes(args.size, args.min_size)
        if callback is not None and hasattr(callback, 'set_steps'):
            callback.set_steps(sum(args.iterations[min(i, len(args.iterations) - 1)]
                                   for i in range(len(sizes))))
        output_raw = None
        for i, size in enumerate(reversed(sizes)):
            content_scaled = []
            for image in content_images:
                if image.size != content_images[0].size:
                    raise ValueError('All of the content images must be the same size')
                content_scaled.append(resize_to_fit(image, size, scale_up=True, div=args.div))
            w, h = content_scaled[0].size
            print('\nScale %d, image size %dx%d.\n' % (i + 1, w, h))
            style_scaled = []
            for image in style_images:
                if args.style_multiscale:
                    style_scaled.append(image)
                elif args.style_scale >= 32:
                    style_scaled.append(resize_to_fit(image, args.style_scale, scale_up=True,
                                                      div=args.div))
                else:
                    style_size = round(size * args.style_scale)
                    if args.max_style_size is not None:
                        style_size = min(style_size, args.max_style_size)
                    style_scaled.append(resize_to_fit(image, style_size,
                                                      scale_up=args.style_scale_up, div=args.div))
            if aux_image:
                aux_scaled = aux_image.resize(content_scaled[0].size, Image.LANCZOS)
                if self.aux_image is not None:
                    self.aux_image.free()
                self.aux_image = self.engine.to_device(self.pil_to_image(aux_scaled))
            if output_raw is not None:      # not the first scale: upsample the averaged iterate
                # model.resize_image (style_transfer.py:399-401): Lanczos, on the GPU
                self.img = resample_device(self.engine, output_raw, (h, w))
                self.optimizer.set_params(self.img)
            else:
                biased_g1 = True
                if initial_image:
                    initial_image = initial_image.resize(content_scaled[0].size, Image.LANCZOS)
                    start = self.pil_to_image(initial_image)
                else:
                    start = self.pil_to_image(np.random.uniform(0, 255, size=(h, w, 3)))
                    biased_g1 = False
                self.img = self.engine.to_device(start)
                if args.optimizer == 'adam':
                    self.optimizer = AdamOptimizer(
                        self.engine, self.img, step_size=args.step_size,
                        bp1=1 - (1 / args.avg_window), decay=args.step_decay[0],
                        power=args.step_decay[1], biased_g1=biased_g1)
                elif args.optimizer == 'lbfgs':
                    self.optimizer = LBFGSOptimizer(self.engine, self.img)
                else:
                    raise ValueError(args.optimizer)
            iters_i = args.iterations[min(i, len(args.iterations) - 1)]
            output_raw = self.transfer(iters_i, content_scaled, style_scaled, callback)
        return self.current_output
