"""Command-line / config-file options, compatible with the reference's ``config_system.py``.

Same flag names, short aliases, types and defaults (``config_system.py:46-119``), same
precedence -- built-in defaults < ``config.py`` next to the CLI script < flags given on the
command line < the file named by ``--config`` (``config_system.py:121-135``) -- and the same
"lazy" values: a config file is executed as Python and may bind an option to a callable of the
run state (``scale``, ``step``, ``steps``, ``img_size``), which is then re-evaluated on every
read (``config_system.py:151-163``).  ``detect_devices()`` asks the HIP runtime instead of
``nvidia-smi`` (``config_system.py:17-24``).
"""

import argparse
from fractions import Fraction
import math
import os
from pathlib import Path
import sys

import numpy as np


def detect_devices():
    """List of visible GPU indices, or [-1] when there is none (the reference's CPU marker)."""
    try:
        from . import lib
        n = lib.device_count()
    except Exception:  # pylint: disable=broad-except
        n = 0
    return list(range(n)) if n else [-1]


def ffloat(text):
    """Floats written as decimals or fractions ('1/3')."""
    return float(Fraction(text))


# (flags, argparse keyword arguments)
_OPTIONS = [
    (('--content-image', '-ci'), dict(help='content image file')),
    (('--style-images', '-si'), dict(nargs='+', default=[], metavar='STYLE_IMAGE',
                                     help='one or more style image files')),
    (('--output-image', '-oi'), dict(help='where to write the result')),
    (('--init-image', '-ii'), dict(metavar='IMAGE', help='start from this image')),
    (('--aux-image', '-ai'), dict(metavar='IMAGE', help='auxiliary image to stay close to')),
    (('--config',), dict(type=Path, help='Python file with option assignments')),
    (('--list-layers',), dict(action='store_true', help='print the model layers and exit')),
    (('--caffe-path',), dict(help='accepted for compatibility; unused (no Caffe involved)')),
    (('--devices',), dict(nargs='+', metavar='DEVICE', type=int, default=[-1],
                          help='GPU indices to farm tiles over (-1: first GPU)')),
    (('--iterations', '-i'), dict(nargs='+', type=int, default=[200, 100],
                                  help='iterations per scale (last value repeats)')),
    (('--size', '-s'), dict(type=int, default=256, help='output size (long edge)')),
    (('--min-size',), dict(type=int, default=182, help='smallest scale of the pyramid')),
    (('--style-scale', '-ss'), dict(type=ffloat, default=1, help='style size relative to content')),
    (('--max-style-size',), dict(type=int, help='upper bound for the style size')),
    (('--style-scale-up',), dict(default=False, action='store_true',
                                 help='allow enlarging style images')),
    (('--style-multiscale', '-sm'), dict(type=int, nargs=2, metavar=('MIN_SCALE', 'MAX_SCALE'),
                                         default=None, help='pool style Grams over these scales')),
    (('--tile-size',), dict(type=int, default=512, help='largest tile edge evaluated at once')),
    (('--optimizer', '-o'), dict(default='adam', choices=['adam', 'lbfgs'], help='optimizer')),
    (('--step-size', '-st'), dict(type=ffloat, default=15, help='Adam step size')),
    (('--step-decay', '-sd'), dict(nargs=2, metavar=('DECAY', 'POWER'), type=ffloat,
                                   default=[0.05, 0.5], help='step size / (1 + DECAY*i)^POWER')),
    (('--avg-window',), dict(type=ffloat, default=20, help='iterate-averaging window')),
    (('--layer-weights',), dict(help='JSON file of per-layer weight factors')),
    (('--content-weight', '-cw'), dict(type=ffloat, default=0.05, help='content factor')),
    (('--dd-weight', '-dw'), dict(type=ffloat, default=0, help='Deep Dream factor')),
    (('--tv-weight', '-tw'), dict(type=ffloat, default=5, help='total-variation factor')),
    (('--tv-power', '-tp'), dict(metavar='BETA', type=ffloat, default=2, help='TV exponent')),
    (('--swt-weight', '-ww'), dict(metavar='WEIGHT', type=ffloat, default=0, help='SWT factor')),
    (('--swt-wavelet', '-wt'), dict(metavar='WAVELET', default='haar', help='SWT wavelet')),
    (('--swt-levels', '-wl'), dict(metavar='LEVELS', default=1, type=int, help='SWT levels')),
    (('--swt-power', '-wp'), dict(metavar='P', default=2, type=ffloat, help='SWT exponent')),
    (('--p-weight', '-pw'), dict(type=ffloat, default=2, help='p-norm factor')),
    (('--p-power', '-pp'), dict(metavar='P', type=ffloat, default=6, help='p-norm exponent')),
    (('--aux-weight', '-aw'), dict(type=ffloat, default=10, help='auxiliary image factor')),
    (('--content-layers',), dict(nargs='*', default=['conv4_2'], metavar='LAYER',
                                 help='content layers (name or name:weight)')),
    (('--style-layers',), dict(nargs='*', metavar='LAYER',
                               default=['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1'],
                               help='style layers (name or name:weight)')),
    (('--dd-layers',), dict(nargs='*', metavar='LAYER', default=[], help='Deep Dream layers')),
    (('--port', '-p'), dict(type=int, default=8000, help='accepted for compatibility')),
    (('--display',), dict(default='browser', choices=['browser', 'gui', 'none'],
                          help='accepted for compatibility (no live view)')),
    (('--browser',), dict(default=None, help='accepted for compatibility')),
    (('--model',), dict(default='vgg19.prototxt', help='deploy prototxt or a stock model name')),
    (('--weights',), dict(default='vgg19.caffemodel', help='.caffemodel / .npz weights')),
    (('--mean',), dict(nargs=3, metavar=('B_MEAN', 'G_MEAN', 'R_MEAN'),
                       default=(103.939, 116.779, 123.68), help='per-channel mean, BGR')),
    (('--save-every',), dict(metavar='N', type=int, default=0, help='save every N steps')),
    (('--seed',), dict(type=int, default=0, help='random seed')),
    (('--div',), dict(metavar='FACTOR', type=int, default=1,
                      help='make image sizes divisible by FACTOR')),
    (('--jitter',), dict(action='store_true', help='per-iteration content features (slow)')),
    (('--debug',), dict(action='store_true', help='verbose logging')),
]


def build_parser():
    parser = argparse.ArgumentParser(
        description='Tiled neural style transfer on AMD MI355X.',
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flags, kwargs in _OPTIONS:
        parser.add_argument(*flags, **kwargs)
    return parser


class ValuePlaceholder:
    """What a lazy option reads as while the run state it depends on does not exist yet
    (config_system.py:147-163 of the reference)."""

    def __repr__(self):
        return 'ValuePlaceholder()'


class LazyArgs:
    """Namespace whose callable values are called with the run-state object on every read."""

    def __init__(self, state, **values):
        object.__setattr__(self, 'state_obj', state)
        object.__setattr__(self, 'ns', argparse.Namespace(**values))

    def __getattr__(self, name):
        value = getattr(object.__getattribute__(self, 'ns'), name)
        if callable(value):
            try:
                return value(object.__getattribute__(self, 'state_obj'))
            except AttributeError:
                return ValuePlaceholder()
        return value

    def __setattr__(self, name, value):
        setattr(self.ns, name, value)

    def __iter__(self):
        return iter(vars(self.ns))

    def __contains__(self, key):
        return key in self.ns

    def __repr__(self):
        return 'LazyArgs(%r)' % vars(self.ns)


CONFIG_SCOPE = dict(detect_devices=detect_devices, math=math, np=np)


def eval_config(path):
    """Runs a config file; its top-level assignments become option values."""
    code = compile(Path(path).read_text(), str(path), 'exec')
    scope = {}
    exec(code, dict(CONFIG_SCOPE), scope)  # pylint: disable=exec-used
    return scope


def parse_args(state=None, argv=None, config_py=None):
    """Returns the merged options.  ``config_py`` defaults to ``config.py`` beside the entry
    script of THIS package (the repository's ``style_transfer.py``), like the reference, which
    looks next to its own config_system.py -- never next to whatever launcher started the process
    (pytest, torch.distributed.run, ...).  ``config_py=False`` reads no default file."""
    parser = build_parser()
    defaults = vars(parser.parse_args([]))
    given = vars(parser.parse_args(argv))
    merged = dict(defaults)
    if config_py is None:
        config_py = Path(__file__).resolve().parent.parent / 'config.py'
    if config_py and Path(config_py).exists():
        merged.update(eval_config(config_py))
    merged.update({k: v for k, v in given.items() if defaults[k] != v})
    if given['config']:
        merged.update(eval_config(given['config']))
    args = LazyArgs(state, **merged)
    if args.debug:
        os.environ['DEBUG'] = '1'
    if not args.list_layers and (not args.content_image or not args.style_images):
        parser.print_help()
        sys.exit(1)
    return args
