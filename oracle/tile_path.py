"""Numpy restatement of the reference's per-tile path and the tile farm around it.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows crowsonkb/style_transfer
``style_transfer.py``: ``eval_features_tile`` 421-427, ``eval_features_once`` 429-464,
``prepare_features`` 466-486, ``eval_sc_grad_tile`` 556-612, ``eval_sc_grad`` 614-645,
``roll``/``roll_features`` 647-661, ``TileWorker.process_one_request`` 230-241,
``eval_loss_and_grad`` 700-736.  Runs on ``oracle.caffe_net.Net``.
"""

import numpy as np

from . import layers as L
from .caffe_net import Net
from .num_ops import (EPS, gram_lower, half_sq_norm, l1_normalize, p_norm_loss_grad, roll_xy,
                      symm_lower_times, tv_loss_grad)


def tile_grid(img_hw, tile_size):
    """Tile rectangles [(y0, y1, x0, x1)] in request order (style_transfer.py:431-451,619-632):
    n = (size-1)//tile+1 tiles per axis of size//n, the last one absorbing the remainder."""
    hw = np.asarray(img_hw)
    n = (hw - 1) // tile_size + 1
    t = hw // n
    rects = []
    for y in range(n[0]):
        for x in range(n[1]):
            y0, x0 = y * t[0], x * t[1]
            y1 = hw[0] if y == n[0] - 1 else y0 + t[0]
            x1 = hw[1] if x == n[1] - 1 else x0 + t[1]
            rects.append((int(y0), int(y1), int(x0), int(x1)))
    return rects


class OracleModel:
    """The reference's ``CaffeModel`` (style_transfer.py:356-661) minus image I/O."""

    def __init__(self, layers, params=None):
        self.net = Net(layers, 1, weights=params)
        self.blob_names = [l['top'] for l in self.net.layers[1:] if l['type'] != 'ReLU']
        # scale = 224 // blob height for a 224 input (layer_info, style_transfer.py:415-419)
        probe = Net(layers, 1, weights=self.net.params)
        probe.blobs['data'].reshape(1, 3, 224, 224)
        probe._reshape()
        self.scale = {b: 224 // probe.blobs[b].data.shape[2] for b in self.blob_names}
        self.channels = {b: probe.blobs[b].data.shape[1] for b in self.blob_names}
        self.last_layer = self.blob_names[-1]
        self.contents = []      # list of {layer: full-image feature map [C, ceil(H/s), ceil(W/s)]}
        self.styles = []        # list of {layer: lower-triangular Gram [C, C]}
        self.img = None

    def deep_to_shallow(self, wanted):
        """Tapped layers ordered deepest first (style_transfer.py:231-233)."""
        return [b for b in reversed(self.blob_names) if b in wanted]

    # ---- style_transfer.py:421-427
    def features_tile(self, tile, wanted):
        net = self.net
        net.blobs['data'].reshape(1, 3, *tile.shape[-2:])
        net.blobs['data'].data[0] = tile
        net.forward(end=self.last_layer)
        np.maximum(net.blobs[self.last_layer].data, 0, out=net.blobs[self.last_layer].data)
        return {b: net.blobs[b].data[0].copy() for b in wanted}

    # ---- style_transfer.py:556-612
    def sc_grad_tile(self, tile, start, content_layers, style_layers, layer_weights,
                     content_weight, style_weight, activations=None, dd_layers=(), dd_weight=None):
        """``activations`` (test-only, see Net.load_activations) replaces the forward results.
        ``dd_layers`` / ``dd_weight``: the Deep-Dream term (style_transfer.py:602-604)."""
        net = self.net
        order = self.deep_to_shallow(list(content_layers) + list(style_layers) + list(dd_layers))
        net.blobs['data'].reshape(1, 3, *tile.shape[-2:])
        net.blobs['data'].data[0] = tile
        net._reshape()
        for b in order:
            net.blobs[b].diff[...] = 0
        net.forward(end=order[0])
        np.maximum(net.blobs[order[0]].data, 0, out=net.blobs[order[0]].data)
        if activations is not None:
            net.load_activations(activations)
        start = np.asarray(start)
        loss = 0.0
        for i, b in enumerate(order):
            lw = layer_weights.get(b, 1.0)
            feat = net.blobs[b].data[0]
            diff = net.blobs[b].diff[0]
            fy, fx = start // self.scale[b]
            fh, fw = feat.shape[-2:]
            if b in content_layers:
                for content in self.contents:
                    resid = feat - content[b][:, fy:fy + fh, fx:fx + fw]
                    loss += lw * content_weight[b] * half_sq_norm(resid)
                    diff += np.float32(lw * content_weight[b]) * l1_normalize(resid)
            if b in style_layers:
                for style in self.styles:
                    gdiff = gram_lower(feat) - style[b]
                    sgrad = symm_lower_times(gdiff, feat.reshape(feat.shape[0], -1))
                    loss += lw * style_weight[b] * half_sq_norm(gdiff) / len(self.styles)
                    diff += np.float32(lw * style_weight[b] / len(self.styles)) * \
                        l1_normalize(sgrad).reshape(feat.shape)
            if b in dd_layers:
                loss -= lw * dd_weight[b] * half_sq_norm(feat)
                diff -= np.float32(lw * dd_weight[b]) * l1_normalize(feat.copy())
            if i + 1 == len(order):
                net.backward(start=b)
            else:
                net.backward(start=b, end=order[i + 1])
        return loss, net.blobs['data'].diff[0].copy()

    # ---- style_transfer.py:647-661 (worker-side roll uses jitter_scale=1, line 234)
    def roll_contents(self, xy_pixels):
        for content in self.contents:
            for b, feat in content.items():
                roll_xy(feat, np.asarray(xy_pixels) // self.scale[b])

    # ---- style_transfer.py:614-645 + worker 230-241
    def sc_grad(self, img, roll, tile_size, content_layers, style_layers, layer_weights,
                content_weight, style_weight):
        loss = 0.0
        grad = np.zeros_like(img)
        for (y0, y1, x0, x1) in tile_grid(img.shape[-2:], tile_size):
            self.roll_contents(roll)
            tl, tg = self.sc_grad_tile(img[:, y0:y1, x0:x1], (y0, x0), content_layers,
                                       style_layers, layer_weights, content_weight, style_weight)
            self.roll_contents(-np.asarray(roll))
            loss += tl
            grad[:, y0:y1, x0:x1] = tg
        return loss, grad

    # ---- style_transfer.py:429-464
    def features_once(self, img, wanted, tile_size):
        hw = np.asarray(img.shape[-2:])
        out = {b: np.zeros((self.channels[b],) + tuple(np.int32(np.ceil(hw / self.scale[b]))),
                           np.float32) for b in wanted}
        for (y0, y1, x0, x1) in tile_grid(hw, tile_size):
            feats = self.features_tile(img[:, y0:y1, x0:x1], wanted)
            for b, f in feats.items():
                fy, fx = y0 // self.scale[b], x0 // self.scale[b]
                out[b][:, fy:fy + f.shape[1], fx:fx + f.shape[2]] = f
        return out

    # ---- style_transfer.py:466-486 (draws from the global numpy RNG exactly like the reference)
    def prepare_features(self, img, wanted, tile_size, passes=10):
        img = img.copy()
        hw = np.asarray(img.shape[-2:])
        if max(hw) <= tile_size:
            passes = 1
        acc = {}
        for i in range(passes):
            xy = np.array((0, 0))
            if i > 0:
                xy = np.int32(np.random.uniform(size=2) * hw) // 32
            roll_xy(img, xy * 32)
            for b in acc:
                roll_xy(acc[b], xy * 32 // self.scale[b])
            feats = self.features_once(img, wanted, tile_size)
            for b in wanted:
                if i == 0:
                    acc[b] = feats[b] / passes
                else:
                    acc[b] += np.float32(1 / passes) * feats[b]
            roll_xy(img, -xy * 32)
            for b in acc:
                roll_xy(acc[b], -xy * 32 // self.scale[b])
        return acc

    def style_grams(self, style_imgs, style_layers, tile_size):
        """Equal-weight mean of the Grams of all style images (style_transfer.py:512-542)."""
        grams, count = {}, 0
        for img in style_imgs:
            feats = self.prepare_features(img, style_layers, tile_size, passes=1)
            for b in feats:
                g = gram_lower(feats[b])
                grams[b] = g if b not in grams else grams[b] + g
            count += 1
        return {b: g / count for b, g in grams.items()}


def regularizer_loss_grad(img, mean, grad, lw_data=1.0, tv_weight=5.0, tv_power=2.0,
                          p_weight=2.0, p_power=6.0, aux_image=None, aux_weight=10.0):
    """Adds TV / p-norm / aux terms to grad in place; returns the added loss
    (style_transfer.py:709-733)."""
    loss = 0.0
    if tv_weight:
        l, g = tv_loss_grad(img / np.float32(127.5), beta=tv_power)
        loss += lw_data * tv_weight * l
        grad += np.float32(lw_data * tv_weight) * g
    if p_weight:
        l, g = p_norm_loss_grad((img + mean - np.float32(127.5)) / np.float32(127.5), p=p_power)
        loss += lw_data * p_weight * l
        grad += np.float32(lw_data * p_weight) * g
    if aux_image is not None:
        a = (img - aux_image) / np.float32(127.5)
        loss += lw_data * aux_weight * half_sq_norm(a)
        grad += np.float32(lw_data * aux_weight) * a
    return loss
