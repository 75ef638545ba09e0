"""Writes seeded synthetic content / style pictures (no photos ship with the reference)."""
import os, sys
import numpy as np
from PIL import Image

def smooth(seed, h, w):
    rng = np.random.RandomState(seed)
    small = rng.uniform(0, 255, (max(2, h // 16), max(2, w // 16), 3)).astype(np.uint8)
    big = np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC), np.float32)
    return np.uint8(np.clip(big + rng.uniform(-16, 16, big.shape), 0, 255))

if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else '.'
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    os.makedirs(out, exist_ok=True)
    Image.fromarray(smooth(0, size, size)).save(os.path.join(out, 'content.png'))
    Image.fromarray(smooth(1, size, size)).save(os.path.join(out, 'style.png'))
    print('wrote', out)
