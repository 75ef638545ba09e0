"""fastpng writes the reference's output file (image.save(path, pnginfo=...),
style_transfer.py:1003-1010) with the deflate spread over threads: any PNG reader must see the
same pixels and the same Comment chunk as from Pillow's writer."""

import io

import numpy as np
from PIL import Image, PngImagePlugin
import pytest

from style_transfer_amd import fastpng


@pytest.mark.parametrize('hw', [(1, 1), (2, 3), (17, 31), (64, 64), (301, 517)])
@pytest.mark.parametrize('threads', [1, 4])
def test_pixels_and_comment_round_trip(hw, threads):
    rng = np.random.RandomState(hw[0] * 7 + threads)
    img = rng.randint(0, 256, hw + (3,)).astype(np.uint8)
    if hw[0] > 16:
        img[hw[0] // 2:] = 77                       # a flat half: long matches across rows
    comment = 'Command line: style_transfer.py -ci a.png\n\nParameters:\nns: Namespace(x=1)\nµ: é\n'
    data = fastpng.encode_rgb(img, [('Comment', comment)], threads=threads, band_rows=5)
    back = Image.open(io.BytesIO(data))
    back.load()
    assert back.mode == 'RGB' and back.size == (hw[1], hw[0])
    assert np.array_equal(np.asarray(back), img)
    ref = io.BytesIO()
    info = PngImagePlugin.PngInfo()
    info.add_itxt('Comment', comment)
    Image.fromarray(img).save(ref, 'PNG', pnginfo=info)
    assert Image.open(io.BytesIO(ref.getvalue())).info == back.info


def test_adler_combine_matches_zlib():
    import zlib
    rng = np.random.RandomState(0)
    a, b = rng.bytes(70001), rng.bytes(12345)
    assert fastpng._adler32_combine(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(a + b)
