import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from argparse import Namespace
from PIL import Image
from style_transfer_amd.config_system import parse_args
from style_transfer_amd.farm import TileFarm
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.transfer import StyleTransfer
from style_transfer_amd.weights import synthetic_weights
from style_transfer_amd import image_ops
z = np.load('/root/repo/tests/golden/reference_vectors.npz')
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
