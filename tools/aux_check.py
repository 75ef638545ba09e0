"""--aux-image: the reference rolls the image, not the auxiliary image.  Runs the reference's
fixture (tests/golden, e2e_aux) with the shift passed to the regularizer kernel and without it
and prints the loss error of both (measured: 1.4e-4 vs 4.3e-3, final image 8.5 vs 55)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from argparse import Namespace
from PIL import Image
from style_transfer_amd.config_system import parse_args
from style_transfer_amd.farm import TileFarm
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.transfer import StyleTransfer
from style_transfer_amd.weights import synthetic_weights
from style_transfer_amd import image_ops
z = np.load(os.path.join(REPO, 'tests', 'golden', 'reference_vectors.npz'))
golden = {k: z[k] for k in z.files}
orig = image_ops.regularizers
for mode in ('rolled', 'unrolled'):
    if mode == 'unrolled':
        image_ops.regularizers = lambda *a, **kw: orig(*a, **{**kw, 'aux_roll': None})
    argv = str(golden['e2e_aux.argv']).split()
    state = Namespace(); args = parse_args(state, argv, config_py=False)
    net = builtin_net(args.model)
    farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
    st = StyleTransfer(farm, args, state)
    log = []
    np.random.seed(args.seed)
    st.transfer_multiscale([Image.fromarray(golden['e2e_aux.content_u8'])], [Image.fromarray(golden['e2e_aux.style_u8'])],
                           aux_image=Image.fromarray(golden['e2e_aux.aux_u8']), callback=lambda **kw: log.append(kw['loss']))
    ref = golden['e2e_aux.log'][:, 2]
    print(mode, np.array(log) / ref - 1, np.abs(st.current_raw.get() - golden['e2e_aux.final_raw']).max())
    farm.close()
