"""Network descriptions: built-in VGG-16/19 graphs and a Caffe deploy-prototxt reader.

The reference describes its models with Caffe ``deploy.prototxt`` files
(``vgg16.prototxt``, ``vgg19.prototxt``, ``*_avgpool.prototxt``, ``*_big.prototxt``) and keys
every option on their layer / blob names (``config_system.py:93-107``,
``style_transfer.py:1030-1073,1098-1101``).  Those names are the compatibility surface, so
this module produces the same graphs two ways:

* ``builtin_net(name)`` builds the six stock variants programmatically (no file needed);
* ``parse_prototxt(text)`` reads any deploy prototxt made of Input / Convolution / ReLU /
  Pooling layers (protobuf text format), for user-supplied ``--model`` files.

A net is a ``NetSpec``: an ordered list of ``LayerSpec`` (name, type, bottom, top and the
conv / pool parameters).  ``to_prototxt`` writes one back out in Caffe's text format.
"""

from dataclasses import dataclass, field
import os
import re


@dataclass
class LayerSpec:
    name: str
    type: str                    # 'Input' | 'Convolution' | 'ReLU' | 'Pooling'
    bottom: str = None
    top: str = None
    num_output: int = 0          # Convolution
    kernel_size: int = 0         # Convolution / Pooling
    pad: int = 0                 # Convolution
    stride: int = 1              # Pooling
    pool: str = 'MAX'            # Pooling: 'MAX' | 'AVE'
    shape: tuple = ()            # Input: (1, 3, H, W)

    def as_dict(self):
        return {k: v for k, v in self.__dict__.items()}


@dataclass
class NetSpec:
    name: str
    layers: list = field(default_factory=list)

    # ---- derived views --------------------------------------------------------------------
    def blob_names(self):
        """Blobs in creation order, without the input blob (``CaffeModel.layers``,
        style_transfer.py:403-413: keys of VGG*_SHAPES)."""
        seen, out = set(), []
        for lay in self.layers[1:]:
            if lay.top not in seen and lay.type != 'ReLU':
                seen.add(lay.top)
                out.append(lay.top)
        return out

    def input_blob(self):
        return self.layers[0].top

    def shapes(self, h=224, w=224):
        """{blob: (C, H, W)} for an h x w input (the VGG*_SHAPES tables for 224 x 224)."""
        dims = {self.input_blob(): (self.layers[0].shape[1] if self.layers[0].shape else 3, h, w)}
        for lay in self.layers[1:]:
            c, bh, bw = dims[lay.bottom]
            if lay.type == 'Convolution':
                o = 2 * lay.pad - lay.kernel_size + 1
                dims[lay.top] = (lay.num_output, bh + o, bw + o)
            elif lay.type == 'Pooling':
                dims[lay.top] = (c, pooled_len(bh, lay.kernel_size, lay.stride),
                                 pooled_len(bw, lay.kernel_size, lay.stride))
        dims.pop(self.input_blob())
        return dims

    def layer_info(self, blob):
        """(scale vs. the image, channels) exactly as ``CaffeModel.layer_info``
        (style_transfer.py:415-419): scale = 224 // blob height at a 224 input."""
        c, h, _ = self.shapes()[blob]
        return 224 // h, c

    def as_dicts(self):
        return [lay.as_dict() for lay in self.layers]


def pooled_len(n, k=2, s=2):
    """Caffe ceil-mode pooled length for pad 0: ceil((n - k) / s) + 1."""
    return max(-((k - n) // s), 0) + 1


# ------------------------------------------------------------------------- built-in graphs ---
_VGG_BLOCKS = {'vgg16': (2, 2, 3, 3, 3), 'vgg19': (2, 2, 4, 4, 4)}
_VGG_WIDTHS = (64, 128, 256, 512, 512)
_VGG_TITLES = {'vgg16': 'VGG_ILSVRC_16_layers', 'vgg19': 'VGG_ILSVRC_19_layers'}


def builtin_net(name):
    """name: 'vgg16' | 'vgg19', optionally suffixed '_avgpool' or '_big', optionally with a
    '.prototxt' extension / directory in front (so ``--model vgg19.prototxt`` resolves here).

    '_avgpool': every pool is AVE instead of MAX.  '_big': conv2_1 reads conv1_2 directly, i.e.
    pool1 is bypassed and stays a dead end (``vgg19_big.prototxt:62``)."""
    stem = os.path.basename(str(name))
    if stem.endswith('.prototxt'):
        stem = stem[:-len('.prototxt')]
    m = re.fullmatch(r'(vgg16|vgg19)(_avgpool|_big)?', stem)
    if not m:
        raise KeyError('no built-in model named %r' % name)
    base, variant = m.group(1), m.group(2)
    net = NetSpec(_VGG_TITLES[base])
    net.layers.append(LayerSpec('input', 'Input', None, 'data', shape=(1, 3, 224, 224)))
    prev = 'data'
    for b, (nconv, width) in enumerate(zip(_VGG_BLOCKS[base], _VGG_WIDTHS), start=1):
        for i in range(1, nconv + 1):
            blob = 'conv%d_%d' % (b, i)
            net.layers.append(LayerSpec(blob, 'Convolution', prev, blob, num_output=width,
                                        kernel_size=3, pad=1))
            net.layers.append(LayerSpec('relu%d_%d' % (b, i), 'ReLU', blob, blob))
            prev = blob
        pool = 'pool%d' % b
        net.layers.append(LayerSpec(pool, 'Pooling', prev, pool, kernel_size=2, stride=2,
                                    pool='AVE' if variant == '_avgpool' else 'MAX'))
        if not (variant == '_big' and b == 1):
            prev = pool
    return net


# ------------------------------------------------------------------------ prototxt reading ---
_TOK = re.compile(r'"(?:[^"\\]|\\.)*"|\'(?:[^\'\\]|\\.)*\'|[{}:]|[^\s{}:"\']+')


def _strip_comments(text):
    return '\n'.join(line.split('#', 1)[0] for line in text.splitlines())


def _parse_message(tokens, i, depth):
    fields = {}
    while i < len(tokens):
        tok = tokens[i]
        if tok == '}':
            if depth == 0:
                raise ValueError('prototxt: unexpected "}"')
            return fields, i + 1
        key, i = tok, i + 1
        if i < len(tokens) and tokens[i] == ':':
            i += 1
        if i >= len(tokens):
            raise ValueError('prototxt: dangling field %r' % key)
        if tokens[i] == '{':
            value, i = _parse_message(tokens, i + 1, depth + 1)
        else:
            value, i = _scalar(tokens[i]), i + 1
        fields.setdefault(key, []).append(value)
    if depth:
        raise ValueError('prototxt: missing "}"')
    return fields, i


def _scalar(tok):
    if tok[0] in '"\'':
        return tok[1:-1]
    try:
        return int(tok)
    except ValueError:
        try:
            return float(tok)
        except ValueError:
            return {'true': True, 'false': False}.get(tok, tok)


def _one(msg, key, default=None):
    vals = msg.get(key)
    return vals[0] if vals else default


def parse_prototxt(text):
    """Reads a Caffe deploy prototxt (new-style ``layer { }`` blocks) into a NetSpec."""
    msg, _ = _parse_message(_TOK.findall(_strip_comments(text)), 0, 0)
    net = NetSpec(_one(msg, 'name', 'net'))
    if 'input' in msg and 'layer' in msg and _one(msg['layer'][0], 'type') != 'Input':
        # legacy header: input: "data"  input_dim: 1 ...  (or input_shape { dim: ... })
        dims = msg.get('input_dim') or _one(msg, 'input_shape', {}).get('dim', [1, 3, 224, 224])
        net.layers.append(LayerSpec('input', 'Input', None, _one(msg, 'input'), shape=tuple(dims)))
    for lay in msg.get('layer', []):
        kind = _one(lay, 'type')
        spec = LayerSpec(_one(lay, 'name'), kind, _one(lay, 'bottom'), _one(lay, 'top'))
        if kind == 'Input':
            shape = _one(_one(lay, 'input_param', {}), 'shape', {})
            spec.shape = tuple(shape.get('dim', [1, 3, 224, 224]))
        elif kind == 'Convolution':
            cp = _one(lay, 'convolution_param', {})
            spec.num_output = _one(cp, 'num_output')
            spec.kernel_size = _one(cp, 'kernel_size', 1)
            spec.pad = _one(cp, 'pad', 0)
            if _one(cp, 'stride', 1) != 1 or _one(cp, 'group', 1) != 1 or \
                    _one(cp, 'dilation', 1) != 1:
                raise ValueError('layer %s: only stride-1 ungrouped convolutions are supported'
                                 % spec.name)
        elif kind == 'Pooling':
            pp = _one(lay, 'pooling_param', {})
            spec.pool = str(_one(pp, 'pool', 'MAX'))
            spec.kernel_size = _one(pp, 'kernel_size')
            spec.stride = _one(pp, 'stride', 1)
            if spec.pool not in ('MAX', 'AVE') or _one(pp, 'pad', 0):
                raise ValueError('layer %s: unsupported pooling parameters' % spec.name)
        elif kind != 'ReLU':
            raise ValueError('layer %s: unsupported layer type %r' % (spec.name, kind))
        net.layers.append(spec)
    if not net.layers or net.layers[0].type != 'Input':
        raise ValueError('prototxt has no Input layer')
    return net


def load_net(model):
    """Resolves ``--model``: an existing prototxt file is parsed; otherwise a stock name
    (``vgg19.prototxt`` ... as shipped by the reference) maps to the built-in graph."""
    if os.path.isfile(str(model)):
        with open(model) as f:
            return parse_prototxt(f.read())
    return builtin_net(model)


def to_prototxt(net):
    """Writes a NetSpec in Caffe's text format (force_backward as the reference's files)."""
    out = ['name: "%s"' % net.name, 'force_backward: true']
    for lay in net.layers:
        out.append('layer {')
        if lay.bottom:
            out.append('  bottom: "%s"' % lay.bottom)
        out += ['  top: "%s"' % lay.top, '  name: "%s"' % lay.name, '  type: "%s"' % lay.type]
        if lay.type == 'Input':
            out += ['  input_param {', '    shape {'] + \
                ['      dim: %d' % d for d in lay.shape] + ['    }', '  }']
        elif lay.type == 'Convolution':
            out += ['  convolution_param {', '    num_output: %d' % lay.num_output,
                    '    pad: %d' % lay.pad, '    kernel_size: %d' % lay.kernel_size, '  }']
        elif lay.type == 'Pooling':
            out += ['  pooling_param {', '    pool: %s' % lay.pool,
                    '    kernel_size: %d' % lay.kernel_size, '    stride: %d' % lay.stride, '  }']
        out.append('}')
    return '\n'.join(out) + '\n'
