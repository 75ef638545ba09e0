"""PNG output without the single-threaded deflate.

The reference ends a run with ``image.save(path, pnginfo=...)`` (``style_transfer.py:1003-1010,
1152-1157``): Pillow's encoder, one zlib stream on one core -- 0.6 s for a 2048 x 2048 picture,
8 % of the whole `--size 2048` run once the GPU part takes 6 s.  A PNG's IDAT data is ONE zlib
stream, but a zlib stream may be assembled from independently compressed pieces: every band of
rows is deflated on its own thread as a raw deflate stream ending on a byte boundary
(``Z_SYNC_FLUSH``; the last band ends the stream), the pieces are concatenated behind one zlib
header and followed by the Adler-32 of the whole filtered image (pigz's construction).  zlib
releases the interpreter lock while it works.  Same pixels, same ``Comment`` text chunk as
Pillow would write; only the compressed bytes differ.
"""

from concurrent.futures import ThreadPoolExecutor
import os
import struct
import zlib

import numpy as np

_SIGNATURE = b'\x89PNG\r\n\x1a\n'


def _chunk(kind, data):
    return struct.pack('>I', len(data)) + kind + data + struct.pack('>I', zlib.crc32(kind + data))


def itxt_chunk(key, text):
    """An uncompressed iTXt chunk, as PngInfo.add_itxt(key, text) writes it."""
    return _chunk(b'iTXt', key.encode('latin-1', 'strict') + b'\0\0\0\0\0' + text.encode('utf-8'))


def _filtered(rows):
    """Filter type 1 (Sub) on every row of an [h, w, 3] uint8 array: bytes [h, 1 + 3w]."""
    h, w, c = rows.shape
    out = np.empty((h, 1 + w * c), np.uint8)
    out[:, 0] = 1
    body = out[:, 1:].reshape(h, w, c)
    body[:, 0] = rows[:, 0]
    np.subtract(rows[:, 1:], rows[:, :-1], out=body[:, 1:])      # modulo 256
    return out


def encode_rgb(rgb, text_chunks=(), level=6, threads=None, band_rows=None):
    """bytes of a PNG file for an [H, W, 3] uint8 array.  text_chunks: [(key, text)] -> iTXt."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3, rgb.shape
    h, w, _ = rgb.shape
    if threads is None:
        threads = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity')
                             else (os.cpu_count() or 1)))
    if band_rows is None:
        band_rows = max(16, -(-h // (threads * 2)))
    bands = [(y, min(h, y + band_rows)) for y in range(0, h, band_rows)]

    def deflate(band):
        y0, y1 = band
        raw = _filtered(rgb[y0:y1]).tobytes()
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
        data = comp.compress(raw)
        data += comp.flush(zlib.Z_FINISH if y1 == h else zlib.Z_SYNC_FLUSH)
        return data, zlib.adler32(raw), len(raw)

    if threads > 1 and len(bands) > 1:
        with ThreadPoolExecutor(max_workers=threads) as pool:
            pieces = list(pool.map(deflate, bands))
    else:
        pieces = [deflate(b) for b in bands]
    adler = 1
    for _, a, n in pieces:
        adler = _adler32_combine(adler, a, n)
    stream = b'\x78\x9c' + b''.join(p[0] for p in pieces) + struct.pack('>I', adler)
    out = [_SIGNATURE, _chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0))]
    out += [itxt_chunk(k, t) for k, t in text_chunks]
    for i in range(0, len(stream), 1 << 20):
        out.append(_chunk(b'IDAT', stream[i:i + (1 << 20)]))
    out.append(_chunk(b'IEND', b''))
    return b''.join(out)


def save_rgb(path, rgb, text_chunks=(), level=6, threads=None):
    with open(path, 'wb') as f:
        f.write(encode_rgb(rgb, text_chunks, level, threads))


def _adler32_combine(adler1, adler2, len2):
    """Adler-32 of A + B from adler32(A), adler32(B) and len(B) (zlib's adler32_combine)."""
    base = 65521
    rem = len2 % base
    sum1 = adler1 & 0xffff
    sum2 = (rem * sum1) % base
    sum1 += (adler2 & 0xffff) + base - 1
    sum2 += ((adler1 >> 16) & 0xffff) + ((adler2 >> 16) & 0xffff) + base - rem
    if sum1 >= base:
        sum1 -= base
    if sum1 >= base:
        sum1 -= base
    if sum2 >= (base << 1):
        sum2 -= (base << 1)
    if sum2 >= base:
        sum2 -= base
    return sum1 | (sum2 << 16)
