"""The e2e_lbfgs fixture under rounding-level switches: how far the final picture moves."""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
import numpy as np
from argparse import Namespace
from PIL import Image
from style_transfer_amd.config_system import parse_args
from style_transfer_amd.farm import TileFarm
from style_transfer_amd.netspec import builtin_net
from style_transfer_amd.transfer import StyleTransfer
from style_transfer_amd.weights import synthetic_weights
golden = np.load(os.path.join(%r, 'tests/golden/reference_vectors.npz'), allow_pickle=False)
argv = str(golden['e2e_lbfgs.argv']).split()
state = Namespace()
args = parse_args(state, argv, config_py=False)
net = builtin_net(args.model)
farm = TileFarm(net, [0], synthetic_weights(net, 0), verbose=False)
st = StyleTransfer(farm, args, state)
log = []
np.random.seed(args.seed)
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    st.transfer_multiscale([Image.fromarray(golden['e2e_lbfgs.content_u8'])],
                       [Image.fromarray(golden['e2e_lbfgs.style0_u8']), Image.fromarray(golden['e2e_lbfgs.style1_u8'])],
                       callback=lambda **kw: log.append((kw['step'], kw['update_size'], kw['loss'], kw['tv_loss'])))
ref, got = golden['e2e_lbfgs.log'], np.float64(log)
diff = np.abs(st.current_raw.get() - golden['e2e_lbfgs.final_raw'])
print(json.dumps({'loss_rel': float(np.abs(got[:, 2] / ref[:, 2] - 1).max()), 'max': float(diff.max()), 'mean': float(diff.mean()),
                  'p999': float(np.percentile(diff, 99.9)), 'over_0.5': int((diff > 0.5).sum()), 'n': int(diff.size)}))
''' % (REPO, REPO)
combos = [{}, {'STX_GRAM': 'bf3', 'STX_SYMM': 'bf3'}, {'STX_GRAM': 'bf3'}, {'STX_SYMM': 'bf3'}, {'STX_GRAM': 'fp32', 'STX_SYMM': 'fp32'},
          {'STX_LBFGS_FUSED': '0'}, {'STX_CONV_H2': '128'}, {'STX_CONV_H2': '0'}, {'STX_CONV_H2': '0', 'STX_GRAM': 'bf3', 'STX_SYMM': 'bf3'},
          {'STX_SUMS_LATE': '0'}, {'STX_GRAM': 'fp32'}, {'STX_SYMM': 'fp32'}]
for c in combos:
    env = dict(os.environ); env.update(c)
    out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print(c, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
