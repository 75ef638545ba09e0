"""A pycaffe-shaped network over the numpy layer restatement.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Implements just the slice of the
pycaffe ``Net`` API that the reference touches (``style_transfer.py:136-146,370,
423-426,559-567,606-610``): ``blobs[name].data/.diff/.reshape()``, ``_layer_names``,
``forward(end=)`` and ``backward(start=, end=)`` with Caffe's inclusive layer-name
ranges, in-place ReLU, overwrite-on-backward bottom diffs and ``force_backward``.

It serves two purposes:
  * ``tests/golden/make_golden.py`` installs it as the module ``caffe`` so that the
    reference's own ``CaffeModel`` code runs unmodified and emits golden vectors;
  * ``oracle/tile_path.py`` (the restatement that travels to the GPU box) runs on it.
"""

import re

import numpy as np

from . import layers as L


# ---------------------------------------------------------------- prototxt (text protobuf) ---
_TOKEN = re.compile(r'\s*(?:#[^\n]*\n\s*)*("(?:[^"\\]|\\.)*"|\'(?:[^\'\\]|\\.)*\'|[{}:]|[^\s{}:#]+)')


def parse_prototxt(text):
    """Parses protobuf text format into nested dicts; every field maps to a list of values."""
    tokens = _TOKEN.findall(text)
    pos = 0

    def scalar(tok):
        if tok[0] in '"\'':
            return tok[1:-1]
        for cast in (int, float):
            try:
                return cast(tok)
            except ValueError:
                pass
        return {'true': True, 'false': False}.get(tok, tok)

    def message(top=False):
        nonlocal pos
        out = {}
        while pos < len(tokens):
            key = tokens[pos]
            if key == '}':
                if top:
                    raise ValueError('unbalanced }')
                pos += 1
                return out
            pos += 1
            if tokens[pos] == ':':
                pos += 1
            if tokens[pos] == '{':
                pos += 1
                value = message()
            else:
                value = scalar(tokens[pos])
                pos += 1
            out.setdefault(key, []).append(value)
        if not top:
            raise ValueError('unterminated message')
        return out

    return message(top=True)


def layers_from_prototxt(text):
    """Returns [{'name','type','bottom','top', ...params}] for the layer types the path uses."""
    net = parse_prototxt(text)
    out = []
    for lay in net.get('layer', []):
        d = dict(name=lay['name'][0], type=lay['type'][0],
                 bottom=(lay.get('bottom') or [None])[0], top=lay['top'][0])
        if d['type'] == 'Input':
            d['shape'] = tuple(lay['input_param'][0]['shape'][0]['dim'])
        elif d['type'] == 'Convolution':
            cp = lay['convolution_param'][0]
            d.update(num_output=cp['num_output'][0], pad=cp.get('pad', [0])[0],
                     kernel_size=cp['kernel_size'][0])
        elif d['type'] == 'Pooling':
            pp = lay['pooling_param'][0]
            d.update(pool=str(pp.get('pool', ['MAX'])[0]), kernel_size=pp['kernel_size'][0],
                     stride=pp.get('stride', [1])[0])
        elif d['type'] != 'ReLU':
            raise ValueError('unsupported layer type %s' % d['type'])
        out.append(d)
    return out


def synthetic_weights(layers, seed=0):
    """Seeded He-style weights N(0, 2/(k*k*Cin)) and biases 0.01*N(0,1)."""
    rng = np.random.RandomState(seed)
    chans = {}
    params = {}
    for lay in layers:
        if lay['type'] == 'Input':
            chans[lay['top']] = lay['shape'][1]
        elif lay['type'] == 'Convolution':
            cin, cout, k = chans[lay['bottom']], lay['num_output'], lay['kernel_size']
            w = rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (k * k * cin))
            b = 0.01 * rng.standard_normal(cout)
            params[lay['name']] = (w.astype(np.float32), b.astype(np.float32))
            chans[lay['top']] = cout
        else:
            chans[lay['top']] = chans[lay['bottom']]
    return params


# ------------------------------------------------------------------------------- the net ---
class Blob:
    """Caffe blob with a leading batch axis of 1: ``data``/``diff`` are [1,C,H,W] float32."""

    def __init__(self, shape):
        self.data = np.zeros(shape, np.float32)
        self.diff = np.zeros(shape, np.float32)

    def reshape(self, *shape):
        if tuple(shape) != self.data.shape:
            self.data = np.zeros(shape, np.float32)
            self.diff = np.zeros(shape, np.float32)


class Net:
    """``Net(deploy, phase, weights=)``: deploy = prototxt path, prototxt text or a layer list;
    weights = {conv name: (w, b)}, a path to an ``.npz`` with ``<name>_w/<name>_b``, or anything
    else (then seeded synthetic weights are used, because no ``.caffemodel`` exists offline)."""

    def __init__(self, deploy, phase=1, weights=None):
        if isinstance(deploy, (list, tuple)):
            self.layers = [dict(l) for l in deploy]
        else:
            text = deploy
            if '\n' not in str(deploy):
                with open(deploy) as f:
                    text = f.read()
            self.layers = layers_from_prototxt(text)
        if isinstance(weights, dict):
            self.params = weights
        elif isinstance(weights, str) and weights.endswith('.npz'):
            z = np.load(weights)
            self.params = {l['name']: (z[l['name'] + '_w'], z[l['name'] + '_b'])
                           for l in self.layers if l['type'] == 'Convolution'}
        else:
            self.params = synthetic_weights(self.layers)
        self._layer_names = [l['name'] for l in self.layers]
        self.blobs = {}
        self._aux = {}
        self._split, self._split_pending = {}, set()     # see _split_consumers
        inp = self.layers[0]
        assert inp['type'] == 'Input'
        self.blobs[inp['top']] = Blob(tuple(inp['shape']))
        self._reshape()

    def _reshape(self):
        """Propagates the input blob's shape through the graph (what Caffe's Reshape does)."""
        for lay in self.layers[1:]:
            c, h, w = self.blobs[lay['bottom']].data.shape[1:]
            if lay['type'] == 'Convolution':
                k, p = lay['kernel_size'], lay['pad']
                shape = (1, lay['num_output'], h + 2 * p - k + 1, w + 2 * p - k + 1)
            elif lay['type'] == 'Pooling':
                k, s = lay['kernel_size'], lay['stride']
                shape = (1, c, L.pooled_size(h, k, s), L.pooled_size(w, k, s))
            else:
                continue
            if lay['top'] not in self.blobs:
                self.blobs[lay['top']] = Blob(shape)
            else:
                self.blobs[lay['top']].reshape(*shape)

    def _index(self, name, default):
        return default if name is None else self._layer_names.index(name)

    def forward(self, start=None, end=None):
        self._reshape()
        i0, i1 = self._index(start, 0), self._index(end, len(self.layers) - 1)
        for lay in self.layers[i0:i1 + 1]:
            t = lay['type']
            if t == 'Input':
                continue
            bottom, top = self.blobs[lay['bottom']], self.blobs[lay['top']]
            if t == 'Convolution':
                w, b = self.params[lay['name']]
                top.data[0] = L.conv_forward(bottom.data[0], w, b, lay['pad'])
            elif t == 'ReLU':
                np.maximum(bottom.data, 0, out=top.data)
            elif t == 'Pooling':
                y, aux = L.pool_forward(bottom.data[0], lay['pool'], lay['kernel_size'],
                                        lay['stride'])
                top.data[0] = y
                self._aux[lay['name']] = aux

    def load_activations(self, data):
        """Overwrites blob data with externally computed activations ({blob: [C,H,W]}) and
        re-derives the MAX-pool argmax from them.  Used by the GPU parity tests to give the
        oracle's backward pass the same discrete decisions (argmax routing, ReLU masks) as
        the implementation under test: those are discontinuous in the activations, so two
        float32 forward passes that agree to 1e-6 may still disagree on a near-tie."""
        for name, arr in data.items():
            self.blobs[name].data[0] = arr
        for lay in self.layers:
            if lay['type'] == 'Pooling' and lay['bottom'] in data:
                _, aux = L.pool_forward(self.blobs[lay['bottom']].data[0], lay['pool'],
                                        lay['kernel_size'], lay['stride'])
                self._aux[lay['name']] = aux

    def _split_consumers(self, blob):
        """Layers that read `blob` without writing it back, if there is more than one: Caffe puts a
        Split layer behind such a blob (net.cpp: InsertSplits) -- every consumer gets a top of its
        own, and on the way back the split ADDS the consumers' diffs into the blob's.  The
        reference's *_big prototxts need it: conv1_2 feeds pool1 (a dead end) and conv2_1
        (vgg19_big.prototxt:62); without the split the later-run pooling layer would overwrite
        conv2_1's gradient with its own zeros."""
        users = [l['name'] for l in self.layers[1:] if l['bottom'] == blob and l['top'] != blob]
        return users if len(users) > 1 else []

    def backward(self, start=None, end=None):
        """Runs layers start..end in reverse (both inclusive, start is the later layer)."""
        i1, i0 = self._index(start, len(self.layers) - 1), self._index(end, 0)
        for lay in reversed(self.layers[i0:i1 + 1]):
            t = lay['type']
            if t == 'Input':
                continue
            bottom, top = self.blobs[lay['bottom']], self.blobs[lay['top']]
            # the Split layer behind a blob with several consumers runs (in reverse order) before
            # the first layer that wrote the blob: diff = sum of what the consumers left
            if lay['top'] in self._split_pending:
                users = self._split_consumers(lay['top'])
                top.diff[...] = sum(self._split.get((lay['top'], u), 0) for u in users)
                self._split_pending.discard(lay['top'])
            if t == 'Convolution':
                w, _ = self.params[lay['name']]
                d = L.conv_backward_data(top.diff[0], w, lay['pad'])
            elif t == 'ReLU':
                bottom.diff[...] = top.diff * (bottom.data > 0)
                continue
            elif t == 'Pooling':
                d = L.pool_backward(top.diff[0], bottom.data.shape[1:], self._aux[lay['name']],
                                    lay['pool'], lay['kernel_size'], lay['stride'])
            else:
                continue
            if self._split_consumers(lay['bottom']):
                self._split[(lay['bottom'], lay['name'])] = d.copy()
                self._split_pending.add(lay['bottom'])
            else:
                bottom.diff[0] = d


# pycaffe module-level functions the reference calls in its worker (style_transfer.py:198-203)
def set_mode_cpu():
    pass


def set_mode_gpu():
    pass


def set_random_seed(_seed):
    pass
