"""The three L-BFGS end-to-end fixtures as runs of the REFERENCE'S OWN Python (build container only).

One place for what make_golden.py sections 4b / 4c / 4j do, so that the sensitivity scripts
(fp32_noise.py, branch_sets.py) run exactly the fixture the golden file holds -- with the reference's
Convolution layer (oracle.layers.conv_forward behind the pycaffe-shaped shim) or its Gram matrix
(num_utils.gram_matrix) optionally replaced by another float32 implementation / a perturbed one.

    cfg4    make_golden 4c: VGG-19 MAX pooling x L-BFGS x ragged 3 x 3 tiles of 30 .. 34 pixels, 3 + 2 iterations
    lbfgs   make_golden 4b: VGG-16 AVE pooling x L-BFGS x two style images x 2 x 2 tiles of 32 .. 40 pixels
    stable  make_golden 4j: VGG-19 MAX pooling x L-BFGS x ragged 2 x 2 tiles of 96 .. 140 pixels (deep planes
            of 12 x 12 and more), 3 + 2 iterations -- the size at which the reference does not branch
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from oracle import layers as L  # noqa: E402

MEAN = (103.939, 116.779, 123.68)
_PLAIN = {}

FIXTURES = {
    'cfg4': dict(argv=['--size', '100', '--min-size', '64', '--tile-size', '40', '--iterations', '3', '2', '-o', 'lbfgs',
                       '--display', 'none', '--seed', '13', '--save-every', '2'],
                 model='vgg19', content=(60, 92, 100), styles=[(61, 80, 70)], progress=True, key='e2e_cfg4'),
    'lbfgs': dict(argv=['--size', '80', '--min-size', '50', '--tile-size', '40', '--iterations', '3', '2', '-o', 'lbfgs',
                        '--model', 'vgg16_avgpool.prototxt', '--display', 'none', '--seed', '9'],
                  model='vgg16_avgpool', content=(50, 64, 80), styles=[(51, 60, 48), (52, 40, 72)], progress=False,
                  key='e2e_lbfgs'),
    'stable': dict(argv=['--size', '280', '--min-size', '190', '--tile-size', '140', '--iterations', '3', '2', '-o', 'lbfgs',
                         '--model', 'vgg19_avgpool.prototxt', '--display', 'none', '--seed', '21'],
                   model='vgg19_avgpool', content=(70, 275, 280), styles=[(71, 200, 230)], progress=False, key='e2e_stable'),
}


def run_fixture(st, config_system, num_utils, which, conv=None, gram=None):
    """One run of fixture `which` through the reference's transfer_multiscale.  conv / gram: replacements
    for the shim's convolution forward pass / the reference's gram_matrix during this run (None: as is).
    Returns (log [steps][4] float64, final raw image, content_u8, [style_u8], argv list)."""
    fx = FIXTURES[which]
    if not _PLAIN:
        _PLAIN.update(conv=L.conv_forward, gram=num_utils.gram_matrix, pool=st.TileWorkerPool)
    L.conv_forward = conv or _PLAIN['conv']
    num_utils.gram_matrix = st.gram_matrix = gram or _PLAIN['gram']
    shapes = st.VGG19_SHAPES if fx['model'].startswith('vgg19') else st.VGG16_SHAPES
    model_args = (os.path.join(mg.REF, fx['model'] + '.prototxt'), 'synthetic', MEAN, shapes)
    names = ['s%d.png' % i for i in range(len(fx['styles']))]
    sys.argv = ['style_transfer.py', '-ci', 'c.png', '-si'] + names + fx['argv']
    st.ARGS = config_system.parse_args(st.STATE)
    st.STATS = st.StatLogger()
    st.STATE.__dict__.clear()
    st.TileWorkerPool = lambda model, devices, caffe_path=None: mg.make_sync_pool(st, model_args, 1, _PLAIN['pool'])
    model = st.CaffeModel(*model_args, placeholder=True)
    transfer = st.StyleTransfer(model)
    content_u8 = mg.smooth_image(*fx['content'])
    styles_u8 = [mg.smooth_image(*s) for s in fx['styles']]
    log = []
    if fx['progress']:
        st.RUN = os.path.join(tempfile.mkdtemp(), 'run')
        progress = st.Progress(transfer, save_every=st.ARGS.save_every,
                               web_if=types.SimpleNamespace(put_event=lambda ev: None), callback=None)

        def cb(**kw):
            log.append((kw['step'], kw['update_size'], kw['loss'], kw['tv_loss']))
            progress(**kw)
        cb.set_steps = progress.set_steps
    else:
        class Cb:
            def set_steps(self, steps):
                self.steps = steps

            def __call__(self, **kw):
                log.append((kw['step'], kw['update_size'], kw['loss'], kw['tv_loss']))
        cb = Cb()
    np.random.seed(st.ARGS.seed)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            transfer.transfer_multiscale([Image.fromarray(content_u8)], [Image.fromarray(s) for s in styles_u8],
                                         None, None, callback=cb)
    finally:
        L.conv_forward = _PLAIN['conv']
        num_utils.gram_matrix = st.gram_matrix = _PLAIN['gram']
    return np.float64(log), transfer.current_raw.copy(), content_u8, styles_u8, sys.argv[1:]
