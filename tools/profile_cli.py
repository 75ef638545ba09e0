"""cProfile of the whole `--size 2048 --tile-size 1024` command-line run (where does the
wall-clock outside the step loop go?)."""
import cProfile, pstats, os, sys, io, contextlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
tmp = tempfile.mkdtemp()
for seed, name in ((0, 'content.png'), (1, 'style.png')):
    r = np.random.RandomState(seed)
    small = r.uniform(0, 255, (128, 128, 3)).astype(np.uint8)
    big = np.asarray(Image.fromarray(small).resize((2048, 2048), Image.BICUBIC), np.float32)
    Image.fromarray(np.uint8(np.clip(big + r.uniform(-16, 16, big.shape), 0, 255))).save(os.path.join(tmp, name))
os.chdir(tmp)
from style_transfer_amd import cli
argv = ['-ci', 'content.png', '-si', 'style.png', '--size', sys.argv[1] if len(sys.argv) > 1 else '2048',
        '--tile-size', '1024', '--weights', 'synthetic', '--display', 'none', '-oi', 'out.png', '--devices', '0']
pr = cProfile.Profile()
out = io.StringIO()
with contextlib.redirect_stdout(out):
    pr.enable(); cli.main(argv); pr.disable()
print([l for l in out.getvalue().splitlines() if 'ending' in l or 'tile-iter' in l])
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
