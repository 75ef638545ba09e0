"""cProfile of BASELINE config 5 (--model vgg16_avgpool.prototxt, two style images, --size 2048)."""
import cProfile, pstats, os, sys, io, contextlib, tempfile, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(REPO, 'tools', 'make_inputs.py'), tmp, '2048'],
                      stdout=subprocess.DEVNULL)
os.chdir(tmp)
from style_transfer_amd import cli
argv = ['-ci', 'content.png', '-si', 'style.png', 'content.png', '--size', '2048', '--tile-size', '1024',
        '--model', 'vgg16_avgpool.prototxt', '--weights', 'synthetic', '--display', 'none', '-oi', 'out.png',
        '--devices', '0']
for rep in range(2):          # (the second run is the warm one)
    pr = cProfile.Profile()
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        pr.enable(); cli.main(argv); pr.disable()
    print('run', rep, [l for l in out.getvalue().splitlines() if 'ending' in l or 'tile-iter' in l])
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45 if rep else 60)
