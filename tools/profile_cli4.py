"""cProfile of BASELINE config 4 (--size 4096 --tile-size 1024 -o lbfgs) on synthetic pictures."""
import cProfile, pstats, os, sys, io, contextlib, tempfile, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(REPO, 'tools', 'make_inputs.py'), tmp, '4096'],
                      stdout=subprocess.DEVNULL)
os.chdir(tmp)
from style_transfer_amd import cli
argv = ['-ci', 'content.png', '-si', 'style.png', '--size', '4096', '--tile-size', '1024', '-o', 'lbfgs',
        '--weights', 'synthetic', '--display', 'none', '-oi', 'out.png', '--devices', '0']
pr = cProfile.Profile()
out = io.StringIO()
with contextlib.redirect_stdout(out):
    pr.enable(); cli.main(argv); pr.disable()
print([l for l in out.getvalue().splitlines() if 'ending' in l or 'tile-iter' in l])
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
