"""The tile farm: cut the image into tiles, evaluate them on the GPUs, stitch the results.

Replaces ``TileWorkerPool`` + ``CaffeModel.eval_features_once / prepare_features /
eval_sc_grad`` (``style_transfer.py:267-337,429-486,614-645``).  The reference forks one process
per device and ships tiles through multiprocessing queues and POSIX shared memory; here ONE
host process drives one ``TileEngine`` per GPU.  Engines only enqueue kernels, so issuing tile
t to engine t mod n (the reference's round-robin, ``style_transfer.py:284-288``) and then
synchronising all engines runs the tiles concurrently.  Tiles are independent -- tile-local
Gram, zero padding at tile borders, disjoint gradient stitch -- so there is no collective:
the only traffic is the tile itself (device-to-device over xGMI when the engine is not the
master GPU) and, once per scale, the targets.

The image lives un-rolled on the master engine; the seam-suppression shift of an iteration is
passed down as ``roll`` and applied as an index offset when tiles are cut and put back.
"""

import os

import numpy as np

from . import image_ops
from .engine import TileEngine


def tile_grid(img_hw, tile_size):
    """[(y0, y1, x0, x1)] in the reference's request order (style_transfer.py:431-451,619-632):
    n = (size - 1) // tile + 1 tiles per axis of size // n; the last row / column absorbs the
    remainder."""
    h, w = int(img_hw[0]), int(img_hw[1])
    ny, nx = (h - 1) // tile_size + 1, (w - 1) // tile_size + 1
    th, tw = h // ny, w // nx
    rects = []
    for y in range(ny):
        for x in range(nx):
            y0, x0 = y * th, x * tw
            rects.append((y0, h if y == ny - 1 else y0 + th, x0, w if x == nx - 1 else x0 + tw))
    return rects


class LazyLoss:
    """A loss whose terms are still on their way from the GPUs: pending tile losses and
    regularizer scalars plus the engines whose ``sync()`` publishes them.  ``float(loss)`` waits
    (for whatever is not finished yet) and adds them up; until then the step loop keeps queueing
    work -- one host synchronisation per optimizer step instead of one per term."""

    def __init__(self):
        self.parts, self.engines, self._value, self._tickets = [], [], None, None

    def add(self, part, engine):
        """part: anything with a ``loss`` or ``value`` attribute that is valid after
        engine.sync()."""
        assert self._value is None
        self.parts.append(part)
        if engine not in self.engines:
            self.engines.append(engine)
        return self

    def seal(self, also=()):
        """Closes the loss behind a fence on every engine that contributes (stx_fence), and on
        the engines in ``also`` (whose other pending values -- step statistics -- are published
        with it): ``float(loss)`` then waits for exactly these terms and not for whatever the
        host has queued since -- the step loop can run one iteration ahead of the GPU."""
        assert self._value is None and self._tickets is None
        for eng in also:
            if eng not in self.engines:
                self.engines.append(eng)
        self._tickets = [(eng, eng.fence()) for eng in self.engines]
        return self

    def __float__(self):
        if self._value is None:
            if self._tickets is not None:
                for eng, ticket in self._tickets:
                    eng.wait_fence(ticket)
            else:
                for eng in self.engines:
                    eng.sync()
            self._value = float(sum(p.loss if hasattr(p, 'loss') else p.value for p in self.parts))
            self.parts, self.engines, self._tickets = [], [], None
        return self._value


class TileFarm:
    """One host process, one engine per entry of ``devices`` (the reference's ``--devices``)."""

    def __init__(self, net, devices=(0,), weights=None, verbose=True, engines=None,
                 streams_per_device=None, force_staging=None):
        """``streams_per_device``: up to this many engines (each with its own HIP stream and
        activation buffers) are created per GPU, lazily, when a step has more tiles than GPUs.
        Tiles of one step then overlap on a GPU, which fills the tails and launch gaps of tiles
        that do not saturate the chip on their own.  Default 2 (STX_STREAMS_PER_GPU overrides): with the
        fp32 kernels of rounds 1-4 four streams were best (4 x 724^2 tiles per step 33.3 -> 27.2 ms,
        1024^2 tiles + 2 %); the fp16-split convolutions fill a launch better and their workgroups
        own a CU each, so two tiles side by side are enough and four only take each other's cache
        -- round 5, alternating on one box: bench.py 248.4 / 248.6 tile-iterations/s with two streams,
        241.8 / 242.3 with four, 237.5 with three, 228.1 with one; BASELINE config 4 13.5 s of
        stepping against 14.1; the four 724^2 tiles of a 1448 scale 9.92 against 9.84 ms."""
        if streams_per_device is None:
            streams_per_device = int(os.environ.get('STX_STREAMS_PER_GPU', '2'))
        self.net = net
        self.verbose = verbose
        # force_staging (or STX_FARM_FORCE_STAGING=1): treat every engine but the master as if it
        # lived on another GPU, i.e. route its tiles through the master-side staging buffers and
        # the cross-engine copies.  On a one-GPU box this runs the multi-GPU leg's code (the
        # copies become device-local); results must not change.
        self.force_staging = bool(int(os.environ.get('STX_FARM_FORCE_STAGING', '0'))) \
            if force_staging is None else bool(force_staging)
        self.owns_engines = engines is None
        self.zero_copy = not bool(int(os.environ.get('STX_FARM_COPY_TILES', '0')))
        self.devices = list(devices)
        self.weights = weights
        self.max_engines = len(self.devices) * max(1, streams_per_device) \
            if engines is None else len(engines)
        # one engine per GPU carries the weights, the packed filter banks and the targets; the
        # other engines of that GPU (more streams, or the same device listed twice) share them
        self.engines = engines if engines is not None else []
        if engines is None:
            for d in self.devices:
                self.engines.append(self._new_engine(d))
        self.master = self.engines[0]
        self._targets = None
        self._tiles = {}        # (engine index, slot, th, tw) -> (tile DeviceArray, grad DeviceArray)
        self._staging = {}      # (slot, th, tw) -> master-side staging for remote engines
        self.tile_evals = 0     # tile-iterations executed (the benchmark's unit of work)

    def _new_engine(self, device):
        for eng in self.engines:
            if eng.device == device:
                return TileEngine(self.net, device, share=eng)
        return TileEngine(self.net, device, self.weights)

    def primaries(self):
        """The engines that own a GPU's shared state (one per distinct group)."""
        out = []
        for eng in self.engines:
            if eng.primary not in out:
                out.append(eng.primary)
        return out

    def close(self):
        for bufs in list(self._tiles.values()) + list(self._staging.values()):
            for b in bufs:
                b.free()
        self._tiles.clear()
        self._staging.clear()
        if self.owns_engines:
            for e in self.engines:
                e.close()

    def layers(self):
        return self.net.blob_names()

    def layer_info(self, layer):
        return self.net.layer_info(layer)

    def set_weights(self, weights):
        self.weights = weights
        for e in self.primaries():
            for name, (w, b) in weights.items():
                e.set_weights(name, w, b)

    def set_contents_and_styles(self, contents, styles):
        """Hands the targets to every GPU, once each (TileWorkerPool.set_contents_and_styles,
        style_transfer.py:309-332, sends them to every worker process)."""
        self._targets = (contents, styles)
        for e in self.engines:          # nothing of the previous scale may still be running
            e.sync()
        for e in self.primaries():
            e.set_contents_and_styles(contents, styles)

    def _engines_for(self, n_tiles):
        """The engines that share a step of n_tiles tiles, creating extra per-GPU engines on
        demand (engine i lives on device i mod n_devices, like the reference's round-robin)."""
        want = min(self.max_engines, max(len(self.devices), n_tiles))
        while len(self.engines) < want and self.owns_engines:
            # (a further stream of a GPU that already has an engine: shares its state)
            self.engines.append(self._new_engine(self.devices[len(self.engines) % len(self.devices)]))
        return self.engines[:want]

    # ------------------------------------------------------------------ eval_features_once
    def eval_features_once(self, img, layers, tile_size=512):
        """Stitched post-ReLU feature maps {layer: [C, ceil(H/s), ceil(W/s)]} of a host image
        [3,H,W] (style_transfer.py:429-464)."""
        img = np.ascontiguousarray(img, np.float32)
        hw = np.array(img.shape[-2:])
        rects = tile_grid(hw, tile_size)
        if len(rects) > 1 and self.verbose:
            nx = (hw[1] - 1) // tile_size + 1
            print('Using %dx%d tiles of size %dx%d.' %
                  (nx, len(rects) // nx, rects[0][3] - rects[0][2], rects[0][1] - rects[0][0]))
        feats = {}
        for layer in layers:
            scale, ch = self.layer_info(layer)
            feats[layer] = np.zeros((ch,) + tuple(np.int32(np.ceil(hw / scale))), np.float32)
        for t, (y0, y1, x0, x1) in enumerate(rects):
            eng = self.engines[t % len(self.engines)]
            tile_feats = eng.features_tile(img[:, y0:y1, x0:x1], list(layers))
            for layer, f in tile_feats.items():
                scale, _ = self.layer_info(layer)
                fy, fx = y0 // scale, x0 // scale
                feats[layer][:, fy:fy + f.shape[1], fx:fx + f.shape[2]] = f
        return feats

    # -------------------------------------------------------------------- prepare_features
    def prepare_features(self, img, layers, tile_size=512, passes=10):
        """Feature maps averaged over randomly rolled tilings to hide the tile seams
        (style_transfer.py:466-486).  Draws from the global numpy RNG like the reference."""
        img = np.array(img, np.float32)
        hw = np.array(img.shape[-2:])
        if max(hw) <= tile_size:
            passes = 1
        feats = {}
        for i in range(passes):
            xy = np.array((0, 0))
            if i > 0:
                xy = np.int32(np.random.uniform(size=2) * hw) // 32
            self._roll_host(img, xy * 32)
            for layer, f in feats.items():
                self._roll_host(f, xy * 32 // self.layer_info(layer)[0])
            once = self.eval_features_once(img, layers, tile_size)
            for layer in layers:
                if i == 0:
                    feats[layer] = once[layer] / passes
                else:
                    feats[layer] += np.float32(1 / passes) * once[layer]
            self._roll_host(img, -xy * 32)
            for layer, f in feats.items():
                self._roll_host(f, -xy * 32 // self.layer_info(layer)[0])
        return feats

    def prepare_features_device(self, img, layers, tile_size=512, passes=10, roll=None):
        """prepare_features with everything on the master GPU: the image is uploaded once, tiles
        are cut with the roll as an index offset, tile maps are stitched by stx_map_place and a
        pass is folded into the average by stx_map_roll_add (acc += roll(feats, -shift) / passes,
        which is the reference's roll-accumulate-unroll of the accumulator, bit for bit).
        Returns {layer: DeviceArray}.  Same RNG draws as the reference.  ``roll``: the maps of
        roll2(img, roll) instead (preprocess_images(..., roll=xy) of the reference's --jitter
        mode, style_transfer.py:789-794); the result stays in that rolled frame."""
        eng = self.master
        img = np.ascontiguousarray(img, np.float32)
        hw = np.array(img.shape[-2:])
        if max(hw) <= tile_size:
            passes = 1
        d_img = eng.to_device(img)
        rects = tile_grid(hw, tile_size)
        if len(rects) > 1 and self.verbose:
            nx = (hw[1] - 1) // tile_size + 1
            print('Using %dx%d tiles of size %dx%d (x %d passes).' %
                  (nx, len(rects) // nx, rects[0][3] - rects[0][2], rects[0][1] - rects[0][0],
                   passes))
        full, acc = {}, {}
        for layer in layers:
            scale, ch = self.layer_info(layer)
            shape = (ch,) + tuple(int(v) for v in np.int32(np.ceil(hw / scale)))
            full[layer] = eng.empty(shape).zero()
            acc[layer] = eng.empty(shape)
        tiles, tile_feats = {}, {}
        for i in range(passes):
            xy = np.array((0, 0))
            if i > 0:
                xy = np.int32(np.random.uniform(size=2) * hw) // 32
            shift = xy * 32
            cut_shift = shift if roll is None else shift + np.asarray(roll, np.int64)
            for rect in rects:
                th, tw = rect[1] - rect[0], rect[3] - rect[2]
                if (th, tw) not in tiles:
                    tiles[(th, tw)] = eng.empty((3, th, tw))
                    tile_feats[(th, tw)] = {}
                tile = tiles[(th, tw)]
                image_ops.cut_tile(eng, d_img, cut_shift, rect, tile)
                feats = eng.features_tile_device(tile, list(layers), tile_feats[(th, tw)])
                for layer in layers:
                    scale, _ = self.layer_info(layer)
                    eng.map_place(full[layer], rect[0] // scale, rect[2] // scale, feats[layer])
            for layer in layers:
                scale, _ = self.layer_info(layer)
                back = -(shift // scale)
                eng.map_roll_add(acc[layer], full[layer], back, 1 / passes,
                                 init_divisor=passes if i == 0 else 0.0)
        eng.sync()
        for bufs in list(tile_feats.values()):
            for b in bufs.values():
                b.free()
        for b in list(tiles.values()) + list(full.values()) + [d_img]:
            b.free()
        return acc

    @staticmethod
    def _roll_host(arr, xy):
        """roll2 (num_utils.py:136-140): xy[0] shifts the last axis, xy[1] the one before."""
        if np.any(np.asarray(xy) != 0):
            arr[...] = np.roll(arr, (int(xy[0]), int(xy[1])), axis=(-1, -2))
        return arr

    def gram_matrix(self, feat):
        return self.master.gram_matrix(feat)

    # ------------------------------------------------------------------------ eval_sc_grad
    def _tile_buffers(self, ei, slot, th, tw):
        key = (ei, slot, th, tw)
        if key not in self._tiles:
            eng = self.engines[ei]
            self._tiles[key] = (eng.empty((3, th, tw)), eng.empty((3, th, tw)))
        return self._tiles[key]

    def _staging_buffers(self, slot, th, tw):
        key = (slot, th, tw)
        if key not in self._staging:
            self._staging[key] = (self.master.empty((3, th, tw)), self.master.empty((3, th, tw)))
        return self._staging[key]

    def eval_sc_grad(self, img, grad, roll, content_layers, style_layers, layer_weights,
                     content_weight, style_weight, tile_size, dd_layers=(), dd_weight=None,
                     content_roll=None, lazy=False):
        """Summed loss and stitched gradient of all tiles (style_transfer.py:614-645).

        img, grad: DeviceArray [3,H,W] on the master GPU, both in the UN-rolled frame; ``roll`` is
        the current iteration's shift in pixels (what the reference passes as the request's
        ``roll`` after physically rolling the image by it).  ``content_roll`` (default: ``roll``)
        is the shift the engines apply to their content maps; --jitter hands them maps that
        are already in the rolled frame and passes (0, 0) (style_transfer.py:789-798).

        Nothing here waits on the host: the master cuts the tiles on its stream, every worker's
        stream waits for the cuts (an event), pulls its tiles, evaluates them and pushes each
        gradient back into a master-side buffer on ITS OWN stream; the master's stream waits for
        each worker's event and stitches.  Workers on other GPUs read and write the master's
        staging buffers directly (xGMI peer copies, all links at once).
        Returns the loss: a float (after synchronising), or with ``lazy`` a LazyLoss and the
        gradient is complete in stream order on the master."""
        if content_roll is None:
            content_roll = roll
        rects = tile_grid(img.shape[-2:], tile_size)
        engines = self._engines_for(len(rects))
        n = len(engines)
        master = self.master
        jobs, workers = [], []
        # every engine evaluates at most one tile this step: the master can cut straight into the
        # engines' input blobs and stitch straight out of their gradient blobs (no copies)
        one_each = len(rects) <= n and self.zero_copy
        for t, rect in enumerate(rects):
            ei, slot = t % n, t // n
            eng = engines[ei]
            th, tw = rect[1] - rect[0], rect[3] - rect[2]
            if eng.device == master.device and not (self.force_staging and ei != 0):
                # the master's own GPU (any stream): its buffers are directly addressable
                tile, tgrad = eng.io_buffers(th, tw) if one_each else self._tile_buffers(ei, slot, th, tw)
            else:
                tile, tgrad = self._staging_buffers(t, th, tw)
            image_ops.cut_tile(master, img, roll, rect, tile)
            if eng is not master and eng not in workers:
                workers.append(eng)
            jobs.append((eng, rect, tile, tgrad))
        for eng in workers:
            eng.wait_for(master)            # the tiles are cut before anyone reads them
        loss = LazyLoss()
        for eng, rect, tile, tgrad in jobs:
            loss.add(eng.sc_grad_tile_async(
                tile, (rect[0], rect[2]), content_roll, content_layers, style_layers,
                layer_weights, content_weight, style_weight, grad_out=tgrad,
                dd_layers=dd_layers, dd_weight=dd_weight), eng)
        for eng in workers:
            master.wait_for(eng)            # that worker's gradients have landed
        for eng, rect, tile, tgrad in jobs:
            image_ops.put_tile(master, grad, roll, rect, tgrad)
        self.tile_evals += len(rects)
        if lazy:
            return loss
        value = float(loss)
        master.sync()
        return value
